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


// ---- decoder side (reference: CGIC/modules/vqvae/decoder.py:304-305,366-378) ---------------------------------
// avgpool_layer1/2 = AvgPool2d(4,4,0) / (2,2,0) on the coarse / medium branch, then inside the up path
//   level -2:  h = h * up2(mask0) + h_medium * mask1                      (medium grid)
//   level -3:  h = h * up4(mask0) + h * up2(mask1) + h_fine * mask2       (fine grid)
// 512 channels at the reference's config: 0.5 GB per tensor at B=64 -- pure HBM streams.  One pass each, float4
// per thread, the reference's products and left-to-right sums (bit-identical; in place is fine: out may alias h).
// The average is the window's row-major running sum divided by the window size, the order of ATen's CPU kernel.

__global__ __launch_bounds__(256) void avgpool_kernel(const float *__restrict__ x, int64_t planes, int64_t H, int64_t W, int k,
                                                      float *__restrict__ out)
{
    const int64_t Ho = H / k, Wo = W / k;
    const int64_t total = planes * Ho * Wo;
    const float div = (float)(k * k);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t xo = t % Wo, r = t / Wo;
        const int64_t yo = r % Ho, pl = r / Ho;
        const float *src = x + (pl * H + yo * k) * W + xo * k;
        float sum = 0.f;
        if (k == 4) {
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const float4 v = *reinterpret_cast<const float4 *>(src + dy * W);     // W % 4 == 0, xo * 4 aligned
                sum += v.x; sum += v.y; sum += v.z; sum += v.w;
            }
        } else {
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const float2 v = *reinterpret_cast<const float2 *>(src + dy * W);
                sum += v.x; sum += v.y;
            }
        }
        out[t] = sum / div;
    }
}

// FINE = false: medium grid [B,C,h,w], masks mask0 [B,h/2,w/2], mask1 [B,h,w]
// FINE = true : fine grid   [B,C,h,w], masks mask0 [B,h/4,w/4], mask1 [B,h/2,w/2], mask2 [B,h,w]
template <bool FINE>
__global__ __launch_bounds__(256) void decoder_blend_kernel(
    const float *__restrict__ hin, const float *__restrict__ own, const int32_t *__restrict__ m0,
    const int32_t *__restrict__ m1, const int32_t *__restrict__ m2, int64_t B, int C, int64_t h, int64_t w,
    float *__restrict__ out)
{
    const int64_t wq = w >> 2;
    const int64_t total = B * C * h * wq;                       // one thread = 4 consecutive x
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t xq = t % wq, r = t / wq;
        const int64_t y = r % h, bc = r / h;
        const int64_t b = bc / C;
        const int64_t x = xq << 2;
        const float4 hv = *reinterpret_cast<const float4 *>(&hin[(bc * h + y) * w + x]);
        const float4 ov = *reinterpret_cast<const float4 *>(&own[(bc * h + y) * w + x]);
        float4 o;
        if (FINE) {
            const float a = (float)m0[(b * (h >> 2) + (y >> 2)) * (w >> 2) + xq];
            const int2 q1 = *reinterpret_cast<const int2 *>(&m1[(b * (h >> 1) + (y >> 1)) * (w >> 1) + (x >> 1)]);
            const int4 q2 = *reinterpret_cast<const int4 *>(&m2[(b * h + y) * w + x]);
            o.x = (hv.x * a + hv.x * (float)q1.x) + ov.x * (float)q2.x;
            o.y = (hv.y * a + hv.y * (float)q1.x) + ov.y * (float)q2.y;
            o.z = (hv.z * a + hv.z * (float)q1.y) + ov.z * (float)q2.z;
            o.w = (hv.w * a + hv.w * (float)q1.y) + ov.w * (float)q2.w;
        } else {
            const int2 q0 = *reinterpret_cast<const int2 *>(&m0[(b * (h >> 1) + (y >> 1)) * (w >> 1) + (x >> 1)]);
            const int4 q1 = *reinterpret_cast<const int4 *>(&m1[(b * h + y) * w + x]);
            o.x = hv.x * (float)q0.x + ov.x * (float)q1.x;
            o.y = hv.y * (float)q0.x + ov.y * (float)q1.y;
            o.z = hv.z * (float)q0.y + ov.z * (float)q1.z;
            o.w = hv.w * (float)q0.y + ov.w * (float)q1.w;
        }
        *reinterpret_cast<float4 *>(&out[(bc * h + y) * w + x]) = o;
    }
}

// medium blend for widths that are even but not multiples of 4 (a 272-px tile column of the 2K path gives a
// 34-wide medium grid): one thread = 2 consecutive x = one coarse-mask element
__global__ __launch_bounds__(256) void decoder_blend_medium2_kernel(
    const float *__restrict__ hin, const float *__restrict__ own, const int32_t *__restrict__ m0,
    const int32_t *__restrict__ m1, int64_t B, int C, int64_t h, int64_t w, float *__restrict__ out)
{
    const int64_t wh = w >> 1;
    const int64_t total = B * C * h * wh;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t xh = t % wh, r = t / wh;
        const int64_t y = r % h, bc = r / h;
        const int64_t b = bc / C;
        const int64_t x = xh << 1;
        const float2 hv = *reinterpret_cast<const float2 *>(&hin[(bc * h + y) * w + x]);
        const float2 ov = *reinterpret_cast<const float2 *>(&own[(bc * h + y) * w + x]);
        const float q0 = (float)m0[(b * (h >> 1) + (y >> 1)) * wh + xh];
        const int2 q1 = *reinterpret_cast<const int2 *>(&m1[(b * h + y) * w + x]);
        float2 o;
        o.x = hv.x * q0 + ov.x * (float)q1.x;
        o.y = hv.y * q0 + ov.y * (float)q1.y;
        *reinterpret_cast<float2 *>(&out[(bc * h + y) * w + x]) = o;
    }
}

// ---- cgic_cut_tiles: pad + crop of the tiling driver as ONE pass -------------------------------------------------------
// inference_high_resolution.py pads the image to a multiple of 16 (centred zeros, :145-173,:227-228) and crops it tile by
// tile (:236-244).  Here every tile of every image is written straight from the UNPADDED image: a destination element is the
// source pixel it covers, or zero where the tile reaches into the pad.  One thread = 4 consecutive destination pixels of one
// row (tile widths are multiples of 16: 16-byte stores; the source is read element-wise because the centred pad may shift it
// by an odd count).  fp32 [N,3,H,W] -> per tile [N, .., 3, th, tw];  uint8 [N,H,W,3] -> per tile [N, .., th, tw, 3].
constexpr int kCutMaxTiles = 96;
struct CutTile {
    void *dst;               // element (image 0, this tile, channel 0 / row 0)
    int64_t image_stride;    // elements between the same tile of consecutive images
    int y0, x0;              // the tile's origin in UNPADDED source coordinates (negative inside the pad)
    int th, tw;
    unsigned int first;      // first work item (4-pixel unit) of this tile
};
struct CutArgs {
    const void *src;
    int H, W, ntiles;
    unsigned int total;      // work items per image
    CutTile t[kCutMaxTiles];
};

template <bool U8>
__global__ __launch_bounds__(256) void cut_tiles_kernel(CutArgs a)
{
    const int64_t n = blockIdx.y;
    const int H = a.H, W = a.W;
    for (unsigned int item = blockIdx.x * 256u + threadIdx.x; item < a.total; item += gridDim.x * 256u) {
        // which tile: every tile's work items are a multiple of 64 (checked on the host; tiles of the x16 grid are), so the 64
        // consecutive items of a wave share their tile: the search and the tile's descriptor stay on the scalar unit
        const unsigned int wbase = __builtin_amdgcn_readfirstlane(item);
        int k = 0;
        while (k + 1 < a.ntiles && wbase >= a.t[k + 1].first) ++k;
        const CutTile &t = a.t[k];
        const unsigned int rel = item - t.first, q = (unsigned int)t.tw >> 2;
        if (U8) {
            const unsigned int r = rel / q, c4 = (rel - r * q) * 4;                 // row, first of 4 columns
            const int sy = t.y0 + (int)r;
            const unsigned char *src = (const unsigned char *)a.src + ((n * H + sy) * (int64_t)W) * 3;
            unsigned int w[3] = {0u, 0u, 0u};
            if (sy >= 0 && sy < H) {
                const int sx = t.x0 + (int)c4;
                if (sx >= 0 && sx + 3 < W && ((((uintptr_t)src) + (unsigned int)sx * 3u) & 3u) == 0) {
                    const unsigned int *p = (const unsigned int *)(src + (int64_t)sx * 3);
                    w[0] = p[0]; w[1] = p[1]; w[2] = p[2];
                } else {
#pragma unroll
                    for (int j = 0; j < 12; ++j) {
                        const int sxj = sx + j / 3;
                        const unsigned int v = (sxj >= 0 && sxj < W) ? src[(int64_t)sxj * 3 + j % 3] : 0u;
                        w[j >> 2] |= v << (8 * (j & 3));
                    }
                }
            }
            unsigned int *dst = (unsigned int *)((unsigned char *)t.dst + n * t.image_stride + ((int64_t)r * t.tw + c4) * 3);
            dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2];
        } else {
            const unsigned int per_plane = (unsigned int)t.th * q;
            const unsigned int ch = rel / per_plane, rr = rel - ch * per_plane, r = rr / q, c4 = (rr - r * q) * 4;
            const int sy = t.y0 + (int)r;
            const float *src = (const float *)a.src + ((n * 3 + ch) * (int64_t)H + sy) * W;
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (sy >= 0 && sy < H) {
                const int sx = t.x0 + (int)c4;
                if (sx >= 0 && sx + 3 < W) {
                    if ((((uintptr_t)(src + sx)) & 15u) == 0) v = *reinterpret_cast<const float4 *>(src + sx);
                    else { v.x = src[sx]; v.y = src[sx + 1]; v.z = src[sx + 2]; v.w = src[sx + 3]; }
                } else {
                    if (sx >= 0 && sx < W) v.x = src[sx];
                    if (sx + 1 >= 0 && sx + 1 < W) v.y = src[sx + 1];
                    if (sx + 2 >= 0 && sx + 2 < W) v.z = src[sx + 2];
                    if (sx + 3 >= 0 && sx + 3 < W) v.w = src[sx + 3];
                }
            }
            float *dst = (float *)t.dst + n * t.image_stride + ((int64_t)ch * t.th + r) * t.tw + c4;
            *reinterpret_cast<float4 *>(dst) = v;
        }
    }
}

}  // namespace cgic

using namespace cgic;

extern "C" int cgic_grain_merge_f32(const float *h_coarse, const float *h_medium, const float *h_fine,
                                    const int32_t *mask_c, const int32_t *mask_m, const int32_t *mask_f, int64_t B,
                                    int C, int64_t h, int64_t w, float *out, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_grain_merge_f32");
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

static int stream_grid(int64_t total)
{
    int64_t nblk = (total + 255) / 256;
    if (nblk > 16384) nblk = 16384;       // 64 workgroups per CU, grid-stride beyond
    return (int)nblk;
}

extern "C" int cgic_avgpool_f32(const float *x, int64_t planes, int64_t H, int64_t W, int k, float *out, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_avgpool_f32");
    CGIC_REQUIRE(x && out, CGIC_ERR_INVALID, "avgpool: NULL tensor");
    CGIC_REQUIRE(k == 2 || k == 4, CGIC_ERR_UNSUPPORTED, "avgpool: window %d; the decoder uses 4 and 2 (decoder.py:304-305)", k);
    CGIC_REQUIRE(planes >= 0 && H > 0 && W > 0 && H % k == 0 && W % k == 0, CGIC_ERR_INVALID,
                 "avgpool: %lldx%lld is not a multiple of the window", (long long)H, (long long)W);
    const int64_t total = planes * (H / k) * (W / k);
    if (total == 0) return CGIC_OK;
    hipLaunchKernelGGL(avgpool_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, planes, H, W, k, out);
    return launch_check("avgpool_kernel");
}

extern "C" int cgic_decoder_blend_medium_f32(const float *h, const float *h_medium, const int32_t *mask_c, const int32_t *mask_m,
                                             int64_t B, int C, int64_t hh, int64_t ww, float *out, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_decoder_blend_medium_f32");
    CGIC_REQUIRE(h && h_medium && mask_c && mask_m && out, CGIC_ERR_INVALID, "decoder_blend_medium: NULL tensor");
    CGIC_REQUIRE(B >= 0 && C > 0 && hh > 0 && ww > 0 && hh % 2 == 0 && ww % 2 == 0, CGIC_ERR_INVALID,
                 "decoder_blend_medium: medium grid %lldx%lld (need even height and width)", (long long)hh, (long long)ww);
    if (ww % 4 != 0) {
        const int64_t total2 = B * C * hh * (ww >> 1);
        if (total2 == 0) return CGIC_OK;
        hipLaunchKernelGGL(decoder_blend_medium2_kernel, dim3(stream_grid(total2)), dim3(256), 0, (hipStream_t)stream, h, h_medium,
                           mask_c, mask_m, B, C, hh, ww, out);
        return launch_check("decoder_blend_medium2_kernel");
    }
    const int64_t total = B * C * hh * (ww >> 2);
    if (total == 0) return CGIC_OK;
    hipLaunchKernelGGL(decoder_blend_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, h, h_medium, mask_c,
                       mask_m, (const int32_t *)nullptr, B, C, hh, ww, out);
    return launch_check("decoder_blend_kernel<medium>");
}

extern "C" int cgic_decoder_blend_fine_f32(const float *h, const float *h_fine, const int32_t *mask_c, const int32_t *mask_m,
                                           const int32_t *mask_f, int64_t B, int C, int64_t hh, int64_t ww, float *out,
                                           cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_decoder_blend_fine_f32");
    CGIC_REQUIRE(h && h_fine && mask_c && mask_m && mask_f && out, CGIC_ERR_INVALID, "decoder_blend_fine: NULL tensor");
    CGIC_REQUIRE(B >= 0 && C > 0 && hh > 0 && ww > 0 && hh % 4 == 0 && ww % 4 == 0, CGIC_ERR_INVALID,
                 "decoder_blend_fine: fine grid %lldx%lld must be positive multiples of 4", (long long)hh, (long long)ww);
    const int64_t total = B * C * hh * (ww >> 2);
    if (total == 0) return CGIC_OK;
    hipLaunchKernelGGL(decoder_blend_kernel<true>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, h, h_fine, mask_c,
                       mask_m, mask_f, B, C, hh, ww, out);
    return launch_check("decoder_blend_kernel<fine>");
}

extern "C" int cgic_cut_tiles(const void *x, int is_u8, int64_t N, int64_t H, int64_t W, int ntiles, const cgic_tile *tiles,
                              cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_cut_tiles");
    CGIC_REQUIRE(x && tiles, CGIC_ERR_INVALID, "cut_tiles: NULL argument");
    CGIC_REQUIRE(N >= 0 && H > 0 && W > 0 && H < (1 << 30) && W < (1 << 30), CGIC_ERR_INVALID, "cut_tiles: bad image shape");
    CGIC_REQUIRE(ntiles >= 1 && ntiles <= kCutMaxTiles, CGIC_ERR_UNSUPPORTED, "cut_tiles: %d tiles (1..%d)", ntiles, kCutMaxTiles);
    CutArgs a;
    a.src = x; a.H = (int)H; a.W = (int)W; a.ntiles = ntiles;
    uint64_t at = 0;
    for (int k = 0; k < ntiles; ++k) {
        const cgic_tile &t = tiles[k];
        CGIC_REQUIRE(t.dst && t.th > 0 && t.tw > 0 && t.tw % 4 == 0, CGIC_ERR_INVALID, "cut_tiles: tile %d: %dx%d (width must be a positive multiple of 4)", k, t.th, t.tw);
        CGIC_REQUIRE(((uintptr_t)t.dst & (is_u8 ? 3u : 15u)) == 0 && (is_u8 ? t.image_stride % 4 == 0 : t.image_stride % 4 == 0), CGIC_ERR_INVALID,
                     "cut_tiles: tile %d: destination not aligned", k);
        // a tile may reach into the pad, never lie wholly outside the image by more than itself
        CGIC_REQUIRE(t.y0 > -(1 << 30) && t.x0 > -(1 << 30) && t.y0 < (1 << 30) && t.x0 < (1 << 30), CGIC_ERR_INVALID, "cut_tiles: tile %d origin", k);
        a.t[k].dst = t.dst; a.t[k].image_stride = t.image_stride; a.t[k].y0 = t.y0; a.t[k].x0 = t.x0; a.t[k].th = t.th; a.t[k].tw = t.tw;
        a.t[k].first = (unsigned int)at;
        at += (uint64_t)(is_u8 ? 1 : 3) * (uint64_t)t.th * (uint64_t)(t.tw / 4);
        CGIC_REQUIRE(at % 64 == 0, CGIC_ERR_UNSUPPORTED, "cut_tiles: tile %d: th * tw / 4 = %lld must be a multiple of 64 (tiles of the x16 grid are)",
                     k, (long long)t.th * (t.tw / 4));
        CGIC_REQUIRE(at < ((uint64_t)1 << 31), CGIC_ERR_UNSUPPORTED, "cut_tiles: image too large");
    }
    for (int k = ntiles; k < kCutMaxTiles; ++k) a.t[k] = a.t[ntiles - 1];
    a.total = (unsigned int)at;
    if (N == 0 || at == 0) return CGIC_OK;
    CGIC_REQUIRE(N <= 65535, CGIC_ERR_UNSUPPORTED, "cut_tiles: more than 65535 images");
    unsigned int nblk = (unsigned int)((at + 255) / 256);          // one item per thread up to 64 workgroups per CU, grid-stride beyond
    if (nblk > 16384) nblk = 16384;
    const dim3 grid(nblk, (unsigned)N);
    if (is_u8)
        hipLaunchKernelGGL(cut_tiles_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(cut_tiles_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return launch_check("cut_tiles_kernel");
}
