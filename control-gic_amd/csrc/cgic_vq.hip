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
#include "cgic_common.h"

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

// ZT = latent tiles (of 16 vectors) per wave; a wave owns 16*ZT vectors and
// scans all K codes; a block owns 4 * 16 * ZT vectors.
template <int ZT>
__global__ __launch_bounds__(kVqThreads) void vq_mfma_kernel(
    const float *__restrict__ z, int64_t hw, int64_t N, const float *__restrict__ cb, int K,
    int64_t *__restrict__ idx_out, float *__restrict__ zq_out, double *__restrict__ sq_partial,
    unsigned long long *__restrict__ hist)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *cbT = smem;                // [4][K]
    float *ee = smem + 4 * K;         // [K]
    unsigned int *lhist = reinterpret_cast<unsigned int *>(smem + 5 * K);  // [K] (only if hist)

    stage_codebook(cb, K, cbT, ee);
    if (hist)
        for (int k = threadIdx.x; k < K; k += blockDim.x) lhist[k] = 0;
    __syncthreads();

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int j = lane & 15;   // column: which latent vector of the tile
    const int g = lane >> 4;   // B-operand k index / C-row group
    const int64_t wave_base = ((int64_t)blockIdx.x * 4 + wave) * (16 * ZT);

    // B operand: lane holds z[n_j][k=g] for each tile; zz per column.
    float zv[ZT], zz[ZT], best[ZT];
    int bi[ZT];
#pragma unroll
    for (int t = 0; t < ZT; ++t) {
        int64_t n = wave_base + 16 * t + j;
        float v = 0.f;
        if (n < N) {
            int64_t b = n / hw, p = n - b * hw;
            v = z[(b * 4 + g) * hw + p];
        }
        zv[t] = v;
        float z0 = __shfl(v, j, kWave), z1 = __shfl(v, 16 + j, kWave);
        float z2 = __shfl(v, 32 + j, kWave), z3 = __shfl(v, 48 + j, kWave);
        zz[t] = sumsq4(z0, z1, z2, z3);
        best[t] = 0.f;
        bi[t] = 0;
    }

    const int ntile = K >> 4;
    for (int ct = 0; ct < ntile; ++ct) {
        // A operand: A[i = lane&15][k = lane>>4] = e[16*ct + i][k]
        float a = cbT[g * K + 16 * ct + j];
        // C rows held by this lane: codes 16*ct + 4*g + r, r = 0..3
        f32x4 e4 = *reinterpret_cast<const f32x4 *>(&ee[16 * ct + 4 * g]);
        const int code0 = 16 * ct;  // + 4*g + r added at the end
#pragma unroll
        for (int t = 0; t < ZT; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, zv[t], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = zz[t] + e4[r];
                float d = __builtin_fmaf(-2.0f, acc[r], s);   // fl(s - 2*mm), 2*mm exact
                // first code of the scan initialises; NaN never replaces (d < best false)
                bool take = (ct == 0 && r == 0) || d < best[t];
                best[t] = take ? d : best[t];
                bi[t] = take ? (code0 + r) : bi[t];
            }
        }
    }

    // Combine the 4 row groups (lanes j, j+16, j+32, j+48): lexicographic (d, index).
    double sq = 0.0;
#pragma unroll
    for (int t = 0; t < ZT; ++t) {
        float d = best[t];
        int i = bi[t] + 4 * g;
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            float od = __shfl_xor(d, off, kWave);
            int oi = __shfl_xor(i, off, kWave);
            // NaN handling mirrors torch.argmin only for non-NaN inputs
            bool take = od < d || (od == d && oi < i);
            d = take ? od : d;
            i = take ? oi : i;
        }
        bi[t] = i;
        int64_t n = wave_base + 16 * t + j;
        if (n < N) {
            if (zq_out || sq_partial) {
                // this lane owns channel g of vector n
                float e = cbT[g * K + i];
                float diff = e - zv[t];
                if (zq_out) {
                    int64_t b = n / hw, p = n - b * hw;
                    zq_out[(b * 4 + g) * hw + p] = zv[t] + diff;
                }
                sq += (double)diff * (double)diff;
            }
            if (g == 0) {
                if (idx_out) idx_out[n] = (int64_t)i;
                if (hist) atomicAdd(&lhist[i], 1u);
            }
        }
    }

    if (sq_partial) {
        // deterministic block reduction: fixed shuffle tree, then waves in order
        __shared__ double wsum[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, kWave);
        if (lane == 0) wsum[wave] = sq;
        __syncthreads();
        if (threadIdx.x == 0) sq_partial[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
    }
    if (hist) {
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            unsigned int c = lhist[k];
            if (c) atomicAdd(&hist[k], (unsigned long long)c);
        }
    }
}

// Plain-VALU restatement: one latent vector per thread, codebook broadcast from
// LDS.  Independent of the MFMA path; used to cross-check it on hardware.
__global__ __launch_bounds__(kVqThreads) void vq_valu_kernel(
    const float *__restrict__ z, int64_t hw, int64_t N, const float *__restrict__ cb, int K,
    int64_t *__restrict__ idx_out, float *__restrict__ zq_out, double *__restrict__ sq_partial,
    unsigned long long *__restrict__ hist)
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
        if (hist) atomicAdd(&hist[bi], 1ull);
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
        if (threadIdx.x == 0) sq_partial[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
    }
}

// loss = m + beta*m (legacy) with m = fp32(mean) -- quantize.py:85-90.  One
// block, fixed summation order => deterministic.
__global__ void vq_loss_kernel(const double *__restrict__ partial, int nblk, double count, float beta,
                               int legacy, float *__restrict__ loss)
{
    __shared__ double s[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) a += partial[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float m = (float)(s[0] / count);
        *loss = legacy ? (m + beta * m) : (beta * m + m);
    }
}

__global__ void index_hist_kernel(const int64_t *__restrict__ idx, int64_t n, int K,
                                  unsigned long long *__restrict__ hist)
{
    extern __shared__ unsigned int lh[];
    for (int k = threadIdx.x; k < K; k += blockDim.x) lh[k] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t v = idx[i];
        if (v >= 0 && v < K) atomicAdd(&lh[v], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x)
        if (lh[k]) atomicAdd(&hist[k], (unsigned long long)lh[k]);
}

static int vq_check(const float *z, int64_t B, int64_t hw, const float *cb, int K, int e_dim,
                    const float *loss, const void *ws)
{
    CGIC_REQUIRE(z && cb, CGIC_ERR_INVALID, "vq: z and codebook must not be NULL");
    CGIC_REQUIRE(B >= 0 && hw >= 0, CGIC_ERR_INVALID, "vq: negative shape");
    CGIC_REQUIRE(e_dim == 4, CGIC_ERR_UNSUPPORTED,
                 "vq: e_dim=%d; this build implements embed_dim == 4 (config_inference.yaml:8)", e_dim);
    CGIC_REQUIRE(K > 0 && K % 16 == 0 && K <= kVqMaxK, CGIC_ERR_UNSUPPORTED,
                 "vq: K=%d; need K %% 16 == 0 and K <= %d", K, kVqMaxK);
    CGIC_REQUIRE(!loss || ws, CGIC_ERR_INVALID, "vq: loss requested without workspace");
    return CGIC_OK;
}

template <int ZT>
static int launch_mfma(const float *z, int64_t hw, int64_t N, const float *cb, int K, int64_t *idx,
                       float *zq, double *part, unsigned long long *hist, hipStream_t s, int *nblk)
{
    const int64_t per_block = 4 * 16 * ZT;
    *nblk = (int)((N + per_block - 1) / per_block);
    size_t lds = sizeof(float) * (size_t)K * (hist ? 6 : 5);
    hipLaunchKernelGGL(vq_mfma_kernel<ZT>, dim3(*nblk), dim3(kVqThreads), lds, s, z, hw, N, cb, K, idx, zq,
                       part, hist);
    return launch_check("vq_mfma_kernel");
}

}  // namespace cgic

using namespace cgic;

extern "C" size_t cgic_vq_workspace_bytes(int64_t n_vectors)
{
    // one double per block of the smallest tiling (64 vectors per block)
    return sizeof(double) * (size_t)((n_vectors + 63) / 64 + 1);
}

extern "C" int cgic_vq_forward_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K,
                                   int e_dim, float beta, int legacy, int64_t *indices, float *z_q,
                                   float *loss, int64_t *hist, void *workspace, cgic_stream_t stream)
{
    int rc = vq_check(z, B, hw, codebook, K, e_dim, loss, workspace);
    if (rc) return rc;
    const int64_t N = B * hw;
    if (N == 0) return CGIC_OK;
    hipStream_t s = (hipStream_t)stream;
    double *part = loss ? (double *)workspace : nullptr;
    unsigned long long *h = (unsigned long long *)hist;
    int nblk = 0;
    // pick the largest per-wave tile that still leaves >= ~2 waves per SIMD busy
    if (N >= (int64_t)512 * 2048) rc = launch_mfma<8>(z, hw, N, codebook, K, indices, z_q, part, h, s, &nblk);
    else if (N >= (int64_t)256 * 1024) rc = launch_mfma<4>(z, hw, N, codebook, K, indices, z_q, part, h, s, &nblk);
    else if (N >= (int64_t)128 * 512) rc = launch_mfma<2>(z, hw, N, codebook, K, indices, z_q, part, h, s, &nblk);
    else rc = launch_mfma<1>(z, hw, N, codebook, K, indices, z_q, part, h, s, &nblk);
    if (rc) return rc;
    if (loss) {
        hipLaunchKernelGGL(vq_loss_kernel, dim3(1), dim3(256), 0, s, part, nblk, (double)N * 4.0, beta, legacy,
                           loss);
        rc = launch_check("vq_loss_kernel");
    }
    return rc;
}

extern "C" int cgic_vq_forward_valu_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K,
                                        int e_dim, float beta, int legacy, int64_t *indices, float *z_q,
                                        float *loss, int64_t *hist, void *workspace, cgic_stream_t stream)
{
    int rc = vq_check(z, B, hw, codebook, K, e_dim, loss, workspace);
    if (rc) return rc;
    const int64_t N = B * hw;
    if (N == 0) return CGIC_OK;
    hipStream_t s = (hipStream_t)stream;
    double *part = loss ? (double *)workspace : nullptr;
    int nblk = (int)((N + kVqThreads - 1) / kVqThreads);
    size_t lds = sizeof(float) * (size_t)K * 5;
    hipLaunchKernelGGL(vq_valu_kernel, dim3(nblk), dim3(kVqThreads), lds, s, z, hw, N, codebook, K, indices, z_q,
                       part, (unsigned long long *)hist);
    rc = launch_check("vq_valu_kernel");
    if (rc) return rc;
    if (loss) {
        hipLaunchKernelGGL(vq_loss_kernel, dim3(1), dim3(256), 0, s, part, nblk, (double)N * 4.0, beta, legacy,
                           loss);
        rc = launch_check("vq_loss_kernel");
    }
    return rc;
}

extern "C" int cgic_index_histogram(const int64_t *indices, int64_t n, int K, int64_t *hist, cgic_stream_t stream)
{
    CGIC_REQUIRE(indices && hist && K > 0 && K <= 16384 && n >= 0, CGIC_ERR_INVALID, "index_histogram: bad args");
    if (n == 0) return CGIC_OK;
    int nblk = (int)((n + 1023) / 1024);
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(index_hist_kernel, dim3(nblk), dim3(256), sizeof(unsigned int) * (size_t)K,
                       (hipStream_t)stream, indices, n, K, (unsigned long long *)hist);
    return launch_check("index_hist_kernel");
}
