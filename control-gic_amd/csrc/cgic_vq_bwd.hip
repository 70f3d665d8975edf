// cgic_vq_bwd.hip -- backward of VectorQuantize2.forward (CGIC/modules/vqvae/quantize.py:85-93) for training
// (CGIC.training_step, model.py:155-174):
//   z_q  = z + (e[idx] - z).detach()                         => dL/dz += g_zq
//   loss = mean((e.detach() - z)^2) + beta * mean((e - z.detach())^2)   (legacy; the two weights swap otherwise)
//        => dL/dz += g_loss * (-2/n * w_z) * (e - z),   dL/de[k] += g_loss * (2/n * w_e) * sum_{idx = k} (e - z)
// The scatter-add into the codebook is DETERMINISTIC: every workgroup accumulates its range of vectors into a private
// [K,4] table in LDS with 64-bit integer atomics on fixed-point values (integer addition is associative, so the order
// of the atomics does not matter; the quantum is 2^-30 of the range's largest |e - z|, i.e. ~200x finer than fp32's),
// writes the table out as doubles, and a second launch adds the workgroups' tables in workgroup order.
// (torch.index_add_ on a GPU uses fp32 atomics: its result changes from run to run.)
#include "cgic_common.h"

namespace cgic {

constexpr int kBwdThreads = 256;
constexpr int kBwdMaxBlocks = 256;

struct BwdArgs {
    const float *z;
    int64_t hw, N;
    const float *cb;
    int K;
    const int64_t *idx;
    const float *g_zq, *g_loss;
    float coef_z;          // -2/n * w_z  (host float, rounded to fp32 like torch rounds a Python scalar)
    float *g_z;
    double *partial;       // [nblk][K*4] or NULL (no codebook gradient wanted)
    int *status;
};

__global__ __launch_bounds__(kBwdThreads) void vq_backward_kernel(BwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *cbs = reinterpret_cast<float4 *>(smem);                                        // [K] codebook rows
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem + (size_t)a.K * 16);   // [K*4] fixed-point sums
    __shared__ unsigned int s_max;
    const int tid = threadIdx.x;
    const int K = a.K;
    const int64_t hw = a.hw, N = a.N;
    for (int k = tid; k < K; k += kBwdThreads) cbs[k] = reinterpret_cast<const float4 *>(a.cb)[k];
    if (a.partial)
        for (int i = tid; i < 4 * K; i += kBwdThreads) acc[i] = 0ull;
    if (tid == 0) s_max = 0;
    __syncthreads();
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = lo + per < N ? lo + per : N;
    const float gl = a.g_loss ? *a.g_loss : 0.f;
    const float cz = gl * a.coef_z;                                   // g_loss * (-scale * w_z), one fp32 rounding like torch

    // pass 1: dz, and the range's largest |e - z| (fixes the fixed-point quantum of pass 2)
    float dmax = 0.f;
    int bad = 0;
    for (int64_t n = lo + tid; n < hi; n += kBwdThreads) {
        const int64_t b = n / hw, p = n - b * hw;
        int64_t k = a.idx[n];
        if (k < 0 || k >= K) { bad = 1; k = 0; }
        const float4 e = cbs[k];
        const float *zp = a.z + b * 4 * hw + p;
        const float d0 = e.x - zp[0], d1 = e.y - zp[hw], d2 = e.z - zp[2 * hw], d3 = e.w - zp[3 * hw];
        dmax = fmaxf(dmax, fmaxf(fmaxf(fabsf(d0), fabsf(d1)), fmaxf(fabsf(d2), fabsf(d3))));
        if (a.g_z) {
            float *gp = a.g_z + b * 4 * hw + p;
            const float *qp = a.g_zq ? a.g_zq + b * 4 * hw + p : nullptr;
            // g_zq + (g_loss * (-scale * w_z)) * diff : a multiplication and an addition, each rounded (quantize.py:85-93 under autograd)
            const float m0 = cz * d0, m1 = cz * d1, m2 = cz * d2, m3 = cz * d3;
            gp[0] = qp ? qp[0] + m0 : m0;
            gp[hw] = qp ? qp[hw] + m1 : m1;
            gp[2 * hw] = qp ? qp[2 * hw] + m2 : m2;
            gp[3 * hw] = qp ? qp[3 * hw] + m3 : m3;
        }
    }
    if (bad && a.status) *a.status = CGIC_ERR_INVALID;
    if (!a.partial) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, kWave));
    if (lane_id() == 0) atomicMax(&s_max, __float_as_uint(dmax));      // non-negative floats order like their bits
    __syncthreads();
    const float M = __uint_as_float(s_max);
    // quantum 2^(e_M - 30): |diff| / quantum < 2^31, a range of up to 2^31 vectors sums below 2^62.  Non-finite or zero M:
    // any scale (the sums are zero or meaningless; NaN gradients are the caller's)
    const int eM = (int)((__float_as_uint(M) >> 23) & 255u);
    const int q = (eM == 0 || eM == 255) ? 0 : 30 - (eM - 127);

    // pass 2: fixed-point accumulation (the latents come back from L2)
    for (int64_t n = lo + tid; n < hi; n += kBwdThreads) {
        const int64_t b = n / hw, p = n - b * hw;
        int64_t k = a.idx[n];
        if (k < 0 || k >= K) k = 0;
        const float4 e = cbs[k];
        const float *zp = a.z + b * 4 * hw + p;
        const float d[4] = {e.x - zp[0], e.y - zp[hw], e.z - zp[2 * hw], e.w - zp[3 * hw]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long v = __float2ll_rn(ldexpf(d[c], q));
            atomicAdd(&acc[k * 4 + c], (unsigned long long)v);
        }
    }
    __syncthreads();
    double *out = a.partial + (size_t)blockIdx.x * 4 * K;
    for (int i = tid; i < 4 * K; i += kBwdThreads) out[i] = ldexp((double)(long long)acc[i], -q);
}

// g_codebook[k][c] = g_loss * (2/n * w_e) * sum over workgroups (in workgroup order) of their tables
__global__ __launch_bounds__(256) void vq_backward_finish_kernel(const double *__restrict__ partial, int nblk, int n_out,
                                                                  const float *__restrict__ g_loss, float coef_e, float *__restrict__ g_cb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    double s = 0.0;
    for (int w = 0; w < nblk; ++w) s += partial[(size_t)w * n_out + i];
    const float ce = (g_loss ? *g_loss : 0.f) * coef_e;
    g_cb[i] = (float)((double)ce * s);
}

static int bwd_blocks(int64_t N)
{
    int64_t nblk = (N + 2047) / 2048;
    if (nblk > kBwdMaxBlocks) nblk = kBwdMaxBlocks;
    return (int)(nblk < 1 ? 1 : nblk);
}

}  // namespace cgic

using namespace cgic;

extern "C" size_t cgic_vq_backward_workspace_bytes(int64_t n_vectors, int K)
{
    return sizeof(double) * (size_t)bwd_blocks(n_vectors) * 4 * (size_t)K + 16;
}

extern "C" int cgic_vq_backward_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K, int e_dim,
                                    const int64_t *indices, const float *g_zq, const float *g_loss, float beta, int legacy,
                                    float *g_z, float *g_codebook, void *workspace, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_vq_backward_f32");
    CGIC_REQUIRE(z && codebook && indices, CGIC_ERR_INVALID, "vq_backward: z, codebook and indices must not be NULL");
    CGIC_REQUIRE(e_dim == 4 && K > 0 && K <= 2048, CGIC_ERR_UNSUPPORTED, "vq_backward: needs a [K<=2048, 4] codebook (K=%d, e_dim=%d)", K, e_dim);
    CGIC_REQUIRE(B >= 0 && hw >= 0, CGIC_ERR_INVALID, "vq_backward: negative shape");
    CGIC_REQUIRE(!g_codebook || workspace, CGIC_ERR_INVALID, "vq_backward: the codebook gradient needs the workspace");
    const int64_t N = B * hw;
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (g_codebook) CGIC_HIP_TRY(hipMemsetAsync(g_codebook, 0, sizeof(float) * 4 * (size_t)K, s));
        return CGIC_OK;
    }
    // quantize.py:85-90: legacy  -> mean((e.detach()-z)^2) + beta*mean((e-z.detach())^2): w_z = 1, w_e = beta; else swapped
    const double scale = 2.0 / ((double)N * 4.0);
    const double w_z = legacy ? 1.0 : (double)beta, w_e = legacy ? (double)beta : 1.0;
    const int nblk = bwd_blocks(N);
    BwdArgs a;
    a.z = z; a.hw = hw; a.N = N; a.cb = codebook; a.K = K; a.idx = indices; a.g_zq = g_zq; a.g_loss = g_loss;
    a.coef_z = (float)(-scale * w_z); a.g_z = g_z; a.partial = g_codebook ? (double *)workspace : nullptr; a.status = nullptr;
    const size_t lds = (size_t)K * 16 + (g_codebook ? (size_t)K * 32 : 0);
    if (lds > 48 * 1024)
        { int rc_ = ensure_dynamic_lds((const void *)vq_backward_kernel, (size_t)lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(vq_backward_kernel, dim3(nblk), dim3(kBwdThreads), lds, s, a);
    int rc = launch_check("vq_backward_kernel");
    if (rc || !g_codebook) return rc;
    const int n_out = 4 * K;
    hipLaunchKernelGGL(vq_backward_finish_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, (const double *)workspace, nblk, n_out,
                       g_loss, (float)(scale * w_e), g_codebook);
    return launch_check("vq_backward_finish_kernel");
}
