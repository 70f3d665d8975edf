// cgic_merge.hip -- the mask-weighted three-grain latent merge that sits directly in front of the
// quantiser (reference: CGIC/modules/vqvae/vqvae_blocks.py:361-366):
//     h = up4(h_coarse) * up4(mask0) + up2(h_medium) * up2(mask1) + h_fine * mask2
// (nearest-neighbour upsampling, masks are 0/1 int32 converted to float).  The reference materialises
// four upsampled temporaries; this is one pass: read the three feature maps at their own resolution,
// write h once.  HBM-bound elementwise work; the products and the left-to-right sums are the
// reference's fp32 operations in the reference's order, so the result is bit-identical.
#include "cgic_common.h"

namespace cgic {

__global__ __launch_bounds__(256) void grain_merge_kernel(
    const float *__restrict__ hc, const float *__restrict__ hm, const float *__restrict__ hf,
    const int32_t *__restrict__ mc, const int32_t *__restrict__ mm, const int32_t *__restrict__ mf,
    int64_t B, int C, int64_t h, int64_t w, float *__restrict__ out)
{
    const int64_t w4 = w >> 2, h4 = h >> 2, w2 = w >> 1, h2 = h >> 1, wq = w >> 2;
    const int64_t total = B * C * h * wq;                       // one thread = 4 consecutive x
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t xq = t % wq, r = t / wq;
        const int64_t y = r % h, bc = r / h;
        const int64_t b = bc / C;
        const int64_t x = xq << 2;
        const float a = hc[(bc * h4 + (y >> 2)) * w4 + xq];
        const float m0 = (float)mc[(b * h4 + (y >> 2)) * w4 + xq];
        const float2 b2 = *reinterpret_cast<const float2 *>(&hm[(bc * h2 + (y >> 1)) * w2 + (x >> 1)]);
        const int2 m1 = *reinterpret_cast<const int2 *>(&mm[(b * h2 + (y >> 1)) * w2 + (x >> 1)]);
        const float4 c4 = *reinterpret_cast<const float4 *>(&hf[(bc * h + y) * w + x]);
        const int4 m2 = *reinterpret_cast<const int4 *>(&mf[(b * h + y) * w + x]);
        const float am = a * m0;
        const float p0 = b2.x * (float)m1.x, p1 = b2.y * (float)m1.y;
        float4 o;
        o.x = (am + p0) + c4.x * (float)m2.x;
        o.y = (am + p0) + c4.y * (float)m2.y;
        o.z = (am + p1) + c4.z * (float)m2.z;
        o.w = (am + p1) + c4.w * (float)m2.w;
        *reinterpret_cast<float4 *>(&out[(bc * h + y) * w + x]) = o;
    }
}

}  // namespace cgic

using namespace cgic;

extern "C" int cgic_grain_merge_f32(const float *h_coarse, const float *h_medium, const float *h_fine,
                                    const int32_t *mask_c, const int32_t *mask_m, const int32_t *mask_f, int64_t B,
                                    int C, int64_t h, int64_t w, float *out, cgic_stream_t stream)
{
    CGIC_REQUIRE(h_coarse && h_medium && h_fine && mask_c && mask_m && mask_f && out, CGIC_ERR_INVALID, "grain_merge: NULL tensor");
    CGIC_REQUIRE(B >= 0 && C > 0 && h > 0 && w > 0 && h % 4 == 0 && w % 4 == 0, CGIC_ERR_INVALID,
                 "grain_merge: fine grid %lldx%lld must be positive multiples of 4", (long long)h, (long long)w);
    const int64_t total = B * C * h * (w >> 2);
    if (total == 0) return CGIC_OK;
    int nblk = (int)((total + 255) / 256);
    if (nblk > 8192) nblk = 8192;
    hipLaunchKernelGGL(grain_merge_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, h_coarse, h_medium, h_fine,
                       mask_c, mask_m, mask_f, B, C, h, w, out);
    return launch_check("grain_merge_kernel");
}
