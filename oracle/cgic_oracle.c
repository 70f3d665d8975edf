/*
 * cgic_oracle.c -- CPU restatement of Control-GIC's granularity-adaptive VQ +
 * router + entropy-coder hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the checker the HIP path is compared
 * against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  The
 * product package (control-gic_amd/) never imports, links or calls it.
 *
 * Parity status: PINNED -- every function below is checked against outputs of
 * the reference itself, generated in the build container by importing
 * /root/reference (tests/golden/make_golden.py -> tests/golden/ fixtures).
 * The reference has no tests or golden vectors of its own.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference root).  Plain C, no dependencies beyond libc/libm.
 * Build: see oracle/Makefile (-ffp-contract=off: every fp32 rounding below is
 * explicit; fmaf() is the only fused operation).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CGIC_K_MAX 65536

/* ------------------------------------------------------------------------- *
 * A. Vector quantiser: CGIC/modules/vqvae/quantize.py:69-97
 *
 *   z  [B,C,h,w] fp32 NCHW  (quantize.py:70 permutes to b h w c)
 *   cb [K,C]     fp32
 *   d = sum(z^2) + sum(e^2) - 2 * z.e          (quantize.py:73-75)
 *   idx = argmin(d, dim=1), lowest index wins   (quantize.py:78)
 *   z_q = z + (e[idx] - z)                      (quantize.py:83,93)
 *   loss = mean((zq-z)^2) + beta*mean((zq-z)^2) (quantize.py:85-90, legacy)
 *
 * fp32 rounding sequence of the CPU reference (torch 2.10 CPU, MKL sgemm with
 * K=C=4; verified bit-for-bit by tests/golden/make_golden.py):
 *   zz = ((z0*z0 + z1*z1) + z2*z2) + z3*z3        each square rounded
 *   ee = same over the codebook row
 *   mm = fma(z3,e3, fma(z2,e2, fma(z1,e1, z0*e0)))
 *   d  = (zz + ee) - 2*mm                         two roundings (2*mm exact)
 * ------------------------------------------------------------------------- */
static inline float sumsq_row(const float *v, int C, long stride)
{
    float s = v[0] * v[0];
    for (int c = 1; c < C; ++c) {
        float q = v[c * stride] * v[c * stride];
        s = s + q;
    }
    return s;
}

int cgic_oracle_vq(const float *z, long B, int C, long hw, const float *cb, int K,
                   float beta, int legacy, int64_t *idx_out, float *zq_out,
                   float *loss_out, int64_t *hist_out)
{
    if (K <= 0 || K > CGIC_K_MAX || C <= 0) return -1;
    float *ee = (float *)malloc(sizeof(float) * (size_t)K);
    if (!ee) return -2;
    for (int k = 0; k < K; ++k) ee[k] = sumsq_row(cb + (long)k * C, C, 1);
    double sq_acc = 0.0;
    for (long b = 0; b < B; ++b) {
        for (long p = 0; p < hw; ++p) {
            const float *zp = z + b * C * hw + p; /* channel stride = hw */
            float zz = sumsq_row(zp, C, hw);
            float best = 0.f;
            int bi = 0;
            for (int k = 0; k < K; ++k) {
                const float *e = cb + (long)k * C;
                float mm = zp[0] * e[0];
                for (int c = 1; c < C; ++c) mm = fmaf(zp[c * hw], e[c], mm);
                float a = zz + ee[k];
                float t = 2.0f * mm;
                float d = a - t;
                /* torch.argmin: first minimum; a NaN counts as minimal and the
                 * first NaN wins (aten/src/ATen/native/cpu/ReduceOpsKernel) */
                if (k == 0) { best = d; bi = 0; }
                else if (!(best != best) && ((d != d) || d < best)) { best = d; bi = k; }
            }
            long n = b * hw + p;
            if (idx_out) idx_out[n] = bi;
            if (hist_out) hist_out[bi] += 1;
            const float *e = cb + (long)bi * C;
            for (int c = 0; c < C; ++c) {
                float zv = zp[c * hw];
                float diff = e[c] - zv;        /* (z_q - z).detach()  :93 */
                float q = zv + diff;           /* z + (...)           :93 */
                if (zq_out) zq_out[b * C * hw + c * hw + p] = q;
                sq_acc += (double)diff * (double)diff;
            }
        }
    }
    if (loss_out) {
        /* torch.mean's summation tree is a torch internal: the oracle keeps the
         * mean in double and the tests hold loss to 1e-6 relative. */
        double m = sq_acc / ((double)B * (double)hw * (double)C);
        float mf = (float)m;
        float l = legacy ? (mf + beta * mf) : (beta * mf + mf);
        *loss_out = l;
    }
    free(ee);
    return 0;
}

/* The distance row of ONE latent vector against the whole codebook, same rounding sequence as
 * cgic_oracle_vq (quantize.py:73-75).  Test infrastructure for the adversarial near-tie tests: lets a
 * test search codebook perturbations whose reference distances differ by exactly 0, 1, 2 ulp. */
int cgic_oracle_vq_distances(const float *zvec, int C, const float *cb, int K, float *d_out)
{
    if (K <= 0 || K > CGIC_K_MAX || C <= 0) return -1;
    float zz = sumsq_row(zvec, C, 1);
    for (int k = 0; k < K; ++k) {
        const float *e = cb + (long)k * C;
        float ee = sumsq_row(e, C, 1);
        float mm = zvec[0] * e[0];
        for (int c = 1; c < C; ++c) mm = fmaf(zvec[c], e[c], mm);
        float a = zz + ee;
        float t = 2.0f * mm;
        d_out[k] = a - t;
    }
    return 0;
}

/* ------------------------------------------------------------------------- *
 * C'. Entropy maps: CGIC/models/model.py:433-483
 *   gray = 0.2989 R + 0.5870 G + 0.1140 B                      (:471)
 *   per p x p patch, 32 bins linspace(-1,1,32), sigma 0.01:
 *   kernel = exp(-0.5 * ((v - bin)/sigma)^2)                    (:452-454)
 *   pdf = mean over pixels; pdf = pdf / (sum(pdf)+1e-40) + 1e-40 (:456-458)
 *   entropy = -sum(pdf * log(pdf))                              (:459)
 * exp/log come from libm here and SLEEF in torch; summation order of
 * torch.mean/sum is internal => this function is held to a tolerance
 * (tests: 2e-5 abs), not bit-exactness.  `bins` is passed in so that the
 * caller supplies torch.linspace's exact fp32 values.
 * ------------------------------------------------------------------------- */
int cgic_oracle_entropy(const float *x, long B, long H, long W, int p,
                        const float *bins, int nbins, float sigma, float *out)
{
    if (p <= 0 || nbins <= 0 || nbins > 256) return -1;
    long hn = H / p, wn = W / p;
    const float eps = 1e-40f;
    for (long b = 0; b < B; ++b) {
        const float *R = x + (b * 3 + 0) * H * W;
        const float *G = x + (b * 3 + 1) * H * W;
        const float *Bc = x + (b * 3 + 2) * H * W;
        for (long py = 0; py < hn; ++py)
            for (long px = 0; px < wn; ++px) {
                float acc[256];
                for (int j = 0; j < nbins; ++j) acc[j] = 0.f;
                for (int iy = 0; iy < p; ++iy)
                    for (int ix = 0; ix < p; ++ix) {
                        long o = (py * p + iy) * W + (px * p + ix);
                        float r = 0.2989f * R[o];
                        float g = 0.5870f * G[o];
                        float bl = 0.1140f * Bc[o];
                        float gray = (r + g) + bl;
                        for (int j = 0; j < nbins; ++j) {
                            float res = gray - bins[j];
                            float t = res / sigma;
                            float t2 = t * t;
                            float a = -0.5f * t2;
                            acc[j] += expf(a);
                        }
                    }
                float norm = 0.f;
                float npix = (float)(p * p);
                for (int j = 0; j < nbins; ++j) { acc[j] = acc[j] / npix; norm += acc[j]; }
                norm += eps;
                float ent = 0.f;
                for (int j = 0; j < nbins; ++j) {
                    float pdf = acc[j] / norm + eps;
                    ent += pdf * logf(pdf);
                }
                out[(b * hn + py) * wn + px] = -ent;
            }
    }
    return 0;
}

/* ---------------------------------------------------------------------------
 * Entropy, REFERENCE ARITHMETIC (round 3): the same quantity as cgic_oracle_entropy, but in torch's CPU operation
 * sequence and summation order, with exp / log correctly rounded (fp64, rounded once) -- the CPU restatement of the
 * opt-in GPU kernel cgic_entropy_maps_ref_f32.  Orders (pinned empirically against torch 2.10's CPU kernels, see
 * tests/golden/make_golden_ties.py and DESIGN.md section 5):
 *   mean over a patch's pixels (model.py:456): cascade sum of an outer reduction -- chunks of 16 consecutive pixels
 *     summed one after the other from 0, the chunk sums added one after the other;
 *   sums over the 32 bins (:457, :459): eight strided partials ((x_k + x_8+k) + x_16+k) + x_24+k, then p_0 + ... + p_7.
 * Against the real Entropy class: 98.6-100 % of the values bit-identical, the rest within 5e-7 (torch's exp / log
 * are MKL's, off the correctly rounded value in ~1 % of the arguments), no mask element flipped on any content family.
 * ------------------------------------------------------------------------- */
static float oracle_sum32_lanes8(const float *x)
{
    float p[8];
    for (int k = 0; k < 8; ++k) p[k] = ((x[k] + x[8 + k]) + x[16 + k]) + x[24 + k];
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s = s + p[k];
    return s;
}

int cgic_oracle_entropy_ref(const float *x, long B, long H, long W, int p,
                            const float *bins, int nbins, float sigma, float *out)
{
    if ((p != 8 && p != 16) || nbins != 32) return -1;
    long hn = H / p, wn = W / p;
    const float eps = 1e-40f;
    const int np = p * p;
    for (long b = 0; b < B; ++b) {
        const float *R = x + (b * 3 + 0) * H * W;
        const float *G = x + (b * 3 + 1) * H * W;
        const float *Bc = x + (b * 3 + 2) * H * W;
        for (long py = 0; py < hn; ++py)
            for (long px = 0; px < wn; ++px) {
                float acc1[32];
                for (int j = 0; j < 32; ++j) acc1[j] = 0.f;
                for (int c = 0; c < np / 16; ++c) {
                    float a0[32];
                    for (int j = 0; j < 32; ++j) a0[j] = 0.f;
                    for (int i = 0; i < 16; ++i) {
                        const int k = 16 * c + i, iy = k / p, ix = k % p;
                        const long o = (py * p + iy) * W + (px * p + ix);
                        const float r = 0.2989f * R[o];
                        const float g = 0.5870f * G[o];
                        const float bl = 0.1140f * Bc[o];
                        const float gray = (r + g) + bl;
                        for (int j = 0; j < 32; ++j) {
                            const float res = gray - bins[j];
                            const float t = res / sigma;
                            const float t2 = t * t;
                            const float a = -0.5f * t2;
                            a0[j] = a0[j] + (float)exp((double)a);
                        }
                    }
                    for (int j = 0; j < 32; ++j) acc1[j] = acc1[j] + a0[j];
                }
                float pdf[32], pl[32];
                for (int j = 0; j < 32; ++j) pdf[j] = acc1[j] / (float)np;
                const float norm = oracle_sum32_lanes8(pdf) + eps;
                for (int j = 0; j < 32; ++j) {
                    const float q = pdf[j] / norm + eps;
                    pl[j] = q * (float)log((double)q);
                }
                out[(b * hn + py) * wn + px] = -oracle_sum32_lanes8(pl);
            }
    }
    return 0;
}

/* The GPU's division by sigma = 0.01f in its reference-arithmetic code (cgic_entropy_dev.h: div_by_sigma001: q = x * 100,
 * corrected once by the exact remainder) against the IEEE quotient the reference computes (model.py:454, `residuals / sigma`):
 * number of fp32 magnitudes u in [lo_bits, hi_bits), stepping by `stride`, for which +x or -x differs in any bit.
 * tests/test_oracle_golden.py samples it; tools/check_fast_div.py runs it exhaustively over 2^-100 <= |x| <= 8 (0 mismatches). */
long cgic_oracle_check_fast_div(unsigned int lo_bits, unsigned int hi_bits, unsigned int stride, unsigned int *first_bad)
{
    const float sigma = 0.01f, r = 100.0f;
    long bad = 0;
    if (stride == 0) stride = 1;
    for (unsigned long long u = lo_bits; u < hi_bits; u += stride) {
        for (int sgn = 0; sgn < 2; ++sgn) {
            const unsigned int bits = (unsigned int)u | (sgn ? 0x80000000u : 0u);
            float x;
            memcpy(&x, &bits, 4);
            const float q = x * r;
            const float q1 = fmaf(fmaf(-q, sigma, x), r, q);
            const float ref = x / sigma;
            if (memcmp(&q1, &ref, 4)) {
                if (!bad && first_bad) *first_bad = bits;
                ++bad;
            }
        }
    }
    return bad;
}

/* ------------------------------------------------------------------------- *
 * C. Router: CGIC/modules/vqvae/RouterTriple.py:8-95
 * ------------------------------------------------------------------------- */
static int cmp_float_asc(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    /* torch.sort ascending: NaN sorts last */
    int xn = x != x, yn = y != y;
    if (xn || yn) return xn - yn;
    return (x > y) - (x < y);
}

/* Python round(): round-half-even on the double (RouterTriple.py:23,30,42,54,65) */
static long py_round(double v) { return (long)nearbyint(v); }

/* mode from the zero-ness of the three ratios (RouterTriple.py:13,19,36,72);
 * fine = 1 - c - m is evaluated in float64 like the Python ctor. */
int cgic_oracle_router_mode(double c, double m)
{
    double f = 1 - c - m;
    int nz = (f == 0) + (m == 0) + (c == 0);
    if (nz == 0) return 0;
    if (nz == 1) return c == 0 ? 1 : (m == 0 ? 2 : 3);
    return c != 0 ? 4 : (m != 0 ? 5 : 6);
}

/* kth-smallest threshold over n values (sorted[k-1], or sorted[0] when k==0) */
static int kth_threshold(const float *v, long n, long k, float *thr)
{
    if (k < 0 || k > n || n <= 0) return -1;
    float *s = (float *)malloc(sizeof(float) * (size_t)n);
    if (!s) return -2;
    memcpy(s, v, sizeof(float) * (size_t)n);
    qsort(s, (size_t)n, sizeof(float), cmp_float_asc);
    *thr = s[k != 0 ? k - 1 : 0];
    free(s);
    return 0;
}

/*
 * e16 [B,h16,w16], e8 [B,2*h16,2*w16].  Thresholds are taken over `nseg`
 * segments of the batch: nseg==1 reproduces the reference (flatten across the
 * batch, RouterTriple.py:21,40,52,63); nseg==B is "B independent B=1 calls".
 * Outputs: mask_c [B,h16,w16], mask_m [B,2h16,2w16], mask_f [B,4h16,4w16] int32;
 * gate (optional) [B,4h16,3*4w16] fp32 = cat(up4(gc), up2(gm), gf) on the last dim.
 */
int cgic_oracle_router(const float *e16, const float *e8, long B, long h16, long w16,
                       double c_ratio, double m_ratio, long nseg,
                       int32_t *mask_c, int32_t *mask_m, int32_t *mask_f, float *gate,
                       int *mode_out)
{
    if (nseg <= 0 || B % nseg != 0) return -1;
    long h8 = 2 * h16, w8 = 2 * w16, h4 = 4 * h16, w4 = 4 * w16;
    long n16 = h16 * w16, n8 = h8 * w8, n4 = h4 * w4;
    int mode = cgic_oracle_router_mode(c_ratio, m_ratio);
    if (mode_out) *mode_out = mode;
    long per = B / nseg;
    int rc = 0;
    float *gc = (float *)calloc((size_t)(B * n16), sizeof(float));
    float *gm = (float *)calloc((size_t)(B * n8), sizeof(float));
    float *gf = (float *)calloc((size_t)(B * n4), sizeof(float));
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(per * n8));
    if (!gc || !gm || !gf || !tmp) { rc = -2; goto done; }

    for (long s = 0; s < nseg && rc == 0; ++s) {
        long b0 = s * per;
        const float *s16 = e16 + b0 * n16;
        const float *s8 = e8 + b0 * n8;
        float *sgc = gc + b0 * n16, *sgm = gm + b0 * n8, *sgf = gf + b0 * n4;
        long N16 = per * n16, N8 = per * n8;
        float thr;
        if (mode == 0 || mode == 2 || mode == 3) {
            long k = py_round((double)N16 * c_ratio);                   /* :23,54,65 */
            if ((rc = kth_threshold(s16, N16, k, &thr))) break;        /* :24 */
            for (long i = 0; i < N16; ++i) sgc[i] = s16[i] < thr ? 1.f : 0.f; /* :25 */
        }
        if (mode == 0) {
            for (long b = 0; b < per; ++b)
                for (long y = 0; y < h8; ++y)
                    for (long x = 0; x < w8; ++x) {
                        float g = sgc[b * n16 + (y / 2) * w16 + (x / 2)];
                        tmp[b * n8 + y * w8 + x] = s8[b * n8 + y * w8 + x] * (1.f - g); /* :27 */
                    }
            /* :30 -- Python evaluates (4*n16)*c + n8*m: int product first, then
             * int*float products in float64, one float64 add */
            long k = py_round((double)(4 * N16) * c_ratio + (double)N8 * m_ratio);
            if ((rc = kth_threshold(tmp, N8, k, &thr))) break;         /* :28,31 */
            for (long b = 0; b < per; ++b)
                for (long y = 0; y < h8; ++y)
                    for (long x = 0; x < w8; ++x) {
                        float g = sgc[b * n16 + (y / 2) * w16 + (x / 2)];
                        long o = b * n8 + y * w8 + x;
                        sgm[o] = (s8[o] < thr ? 1.f : 0.f) * ((1.f - g) != 0.f ? 1.f : 0.f); /* :32 */
                    }
        } else if (mode == 1) {
            long k = py_round((double)N8 * m_ratio);                    /* :42 */
            if ((rc = kth_threshold(s8, N8, k, &thr))) break;
            for (long i = 0; i < N8; ++i) sgm[i] = s8[i] < thr ? 1.f : 0.f; /* :44 */
        } else if (mode == 3) {
            for (long b = 0; b < per; ++b)
                for (long y = 0; y < h8; ++y)
                    for (long x = 0; x < w8; ++x)
                        sgm[b * n8 + y * w8 + x] = 1.f - sgc[b * n16 + (y / 2) * w16 + (x / 2)]; /* :68 */
        } else if (mode == 4) {
            for (long i = 0; i < N16; ++i) sgc[i] = 1.f;               /* :75 */
        } else if (mode == 5) {
            for (long i = 0; i < N8; ++i) sgm[i] = 1.f;                /* :81 */
        }
        /* fine gate */
        for (long b = 0; b < per; ++b)
            for (long y = 0; y < h4; ++y)
                for (long x = 0; x < w4; ++x) {
                    float c4 = sgc[b * n16 + (y / 4) * w16 + (x / 4)];
                    float m2 = sgm[b * n8 + (y / 2) * w8 + (x / 2)];
                    float f;
                    switch (mode) {
                    case 0: f = (1.f - c4) - m2; break;                /* :34 */
                    case 1: f = 1.f - m2; break;                       /* :47 */
                    case 2: f = 1.f - c4; break;                       /* :58 */
                    case 6: f = 1.f; break;                            /* :87 */
                    default: f = 0.f; break;                           /* :69,77,83 */
                    }
                    sgf[b * n4 + y * w4 + x] = f;
                }
    }
    if (rc == 0) {
        /* mask = gate.bool().int()  (RouterTriple.py:92) */
        for (long i = 0; i < B * n16; ++i) mask_c[i] = gc[i] != 0.f;
        for (long i = 0; i < B * n8; ++i) mask_m[i] = gm[i] != 0.f;
        for (long i = 0; i < B * n4; ++i) mask_f[i] = gf[i] != 0.f;
        if (gate)                                                      /* :93 */
            for (long b = 0; b < B; ++b)
                for (long y = 0; y < h4; ++y)
                    for (long x = 0; x < w4; ++x) {
                        float *row = gate + (b * h4 + y) * 3 * w4;
                        row[x] = gc[b * n16 + (y / 4) * w16 + (x / 4)];
                        row[w4 + x] = gm[b * n8 + (y / 2) * w8 + (x / 2)];
                        row[2 * w4 + x] = gf[b * n4 + y * w4 + x];
                    }
    }
done:
    free(gc); free(gm); free(gf); free(tmp);
    return rc;
}

/* ------------------------------------------------------------------------- *
 * D. Huffman table: CGIC/tools/indices_coding.py:10-17,19-27,46-75
 *
 * The tree's shape under frequency ties is decided by CPython's heapq
 * (Lib/heapq.py, stdlib 3.10; the C accelerator _heapq is the same algorithm):
 *   heappush = append + _siftdown(heap, 0, len-1)
 *   heappop  = pop last; if heap non-empty: swap into root, _siftup(heap, 0)
 *   _siftup walks the hole to a leaf promoting the right child unless
 *   left < right, then _siftdown's the displaced item back up.
 * Ordering is HeapNode.__lt__ = freq only (indices_coding.py:26-27).
 * ------------------------------------------------------------------------- */
typedef struct {
    int64_t freq;
    int sym;          /* -1 for merged nodes (char None) */
    int left, right;  /* node ids, -1 = None */
} hnode_t;

static void hq_siftdown(int *heap, const hnode_t *nd, int startpos, int pos)
{
    int newitem = heap[pos];
    while (pos > startpos) {
        int parentpos = (pos - 1) >> 1;
        int parent = heap[parentpos];
        if (nd[newitem].freq < nd[parent].freq) {
            heap[pos] = parent;
            pos = parentpos;
            continue;
        }
        break;
    }
    heap[pos] = newitem;
}

static void hq_siftup(int *heap, const hnode_t *nd, int endpos, int pos)
{
    int startpos = pos;
    int newitem = heap[pos];
    int childpos = 2 * pos + 1;
    while (childpos < endpos) {
        int rightpos = childpos + 1;
        if (rightpos < endpos && !(nd[heap[childpos]].freq < nd[heap[rightpos]].freq))
            childpos = rightpos;
        heap[pos] = heap[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    heap[pos] = newitem;
    hq_siftdown(heap, nd, startpos, pos);
}

static void hq_push(int *heap, int *len, const hnode_t *nd, int item)
{
    heap[(*len)++] = item;
    hq_siftdown(heap, nd, 0, *len - 1);
}

static int hq_pop(int *heap, int *len, const hnode_t *nd)
{
    int last = heap[--(*len)];
    if (*len > 0) {
        int ret = heap[0];
        heap[0] = last;
        hq_siftup(heap, nd, *len, 0);
        return ret;
    }
    return last;
}

/*
 * freq[n] indexed by symbol (already int(value.item())); order[n] = the symbol
 * pushed i-th, i.e. the iteration order of the `frequency` mapping
 * (indices_coding.py:46-49).  NB: the reference's mapping is
 * nn.ParameterDict({str(i): ...}) (quantize.py:28), and ParameterDict.update
 * SORTS a plain dict's keys as strings, so the real push order is
 * '0','1','10','100','1000',...  (order==NULL means 0,1,2,...).
 * Outputs: len_out[n] code length in bits; code_out[n * words] the code bits,
 * MSB-first (bit i of the code is bit 31-(i%32) of word i/32); words must be
 * >= ceil(maxlen/32) -- call once with code_out==NULL to get maxlen.
 * Returns max code length, or <0 on error.
 */
int cgic_oracle_huffman_build(const int64_t *freq, const int32_t *order, int n,
                              int32_t *len_out, uint32_t *code_out, int words)
{
    if (n <= 0 || n > CGIC_K_MAX) return -1;
    int total = 2 * n;
    hnode_t *nd = (hnode_t *)malloc(sizeof(hnode_t) * (size_t)total);
    int *heap = (int *)malloc(sizeof(int) * (size_t)n);
    int hlen = 0, nn = 0;
    for (int i = 0; i < n; ++i) {                       /* make_heap :46-49 */
        int s = order ? order[i] : i;
        if (s < 0 || s >= n) { free(nd); free(heap); return -4; }
        nd[nn].freq = freq[s]; nd[nn].sym = s; nd[nn].left = nd[nn].right = -1;
        hq_push(heap, &hlen, nd, nn++);
    }
    while (hlen > 1) {                                  /* merge_nodes :51-60 */
        int a = hq_pop(heap, &hlen, nd);
        int b = hq_pop(heap, &hlen, nd);
        nd[nn].freq = nd[a].freq + nd[b].freq; nd[nn].sym = -1;
        nd[nn].left = a; nd[nn].right = b;
        hq_push(heap, &hlen, nd, nn++);
    }
    int root = hq_pop(heap, &hlen, nd);                 /* make_codes :73-75 */
    /* make_codes_helper :62-71, iterative DFS; code = path, '0' left '1' right.
     * We carry the path as a bit array of up to n bits. */
    int maxlen = 0;
    int pw = (n + 31) / 32 + 1;
    typedef struct { int node; int depth; } frame_t;
    frame_t *stack = (frame_t *)malloc(sizeof(frame_t) * (size_t)(total + 2));
    uint32_t *paths = (uint32_t *)calloc((size_t)(total + 2) * (size_t)pw, sizeof(uint32_t));
    int sp = 0;
    stack[0].node = root; stack[0].depth = 0; sp = 1;
    while (sp > 0) {
        --sp;
        int node = stack[sp].node, depth = stack[sp].depth;
        uint32_t cur[ (CGIC_K_MAX + 31) / 32 + 1 ];
        memcpy(cur, paths + (size_t)sp * pw, sizeof(uint32_t) * (size_t)pw);
        if (nd[node].sym >= 0) {
            int s = nd[node].sym;
            len_out[s] = depth;
            if (depth > maxlen) maxlen = depth;
            if (code_out) {
                int cw = (depth + 31) / 32;
                if (cw > words) { free(nd); free(heap); free(stack); free(paths); return -3; }
                for (int j = 0; j < words; ++j) code_out[(size_t)s * words + j] = j < cw ? cur[j] : 0;
            }
        }
        /* push right (code+'1') then left (code+'0'); None children are
         * skipped when popped in the reference, so skip them here. */
        if (nd[node].right >= 0) {
            memcpy(paths + (size_t)sp * pw, cur, sizeof(uint32_t) * (size_t)pw);
            paths[(size_t)sp * pw + depth / 32] |= 1u << (31 - depth % 32);
            stack[sp].node = nd[node].right; stack[sp].depth = depth + 1; ++sp;
        }
        if (nd[node].left >= 0) {
            memcpy(paths + (size_t)sp * pw, cur, sizeof(uint32_t) * (size_t)pw);
            stack[sp].node = nd[node].left; stack[sp].depth = depth + 1; ++sp;
        }
    }
    free(nd); free(heap); free(stack); free(paths);
    return maxlen;
}

/* ------------------------------------------------------------------------- *
 * E. Huffman encode: indices_coding.py:78-126 (get_encoded_text, pad_encoded_text,
 *    get_byte_array, compress).  Also E': mask_coding.py:14-55 with the fixed
 *    code {0:'0', 1:'1'} (pass len={1,1}, code={0x00000000,0x80000000}).
 * Returns bytes written (0 for an empty input: empty file, :116-118), <0 error.
 * ------------------------------------------------------------------------- */
long cgic_oracle_encode(const int64_t *syms, long n, const int32_t *len,
                        const uint32_t *code, int words, int nsym,
                        uint8_t *out, long cap)
{
    if (n == 0) return 0;
    long bits = 0;
    for (long i = 0; i < n; ++i) {
        if (syms[i] < 0 || syms[i] >= nsym) return -1;  /* KeyError in the reference */
        bits += len[syms[i]];
    }
    int pad = 8 - (int)(bits % 8);                      /* :92, 1..8 */
    long nbytes = 1 + (bits + pad) / 8;
    if (nbytes > cap) return -2;
    memset(out, 0, (size_t)nbytes);
    out[0] = (uint8_t)pad;                              /* :96-97 */
    long pos = 8;
    for (long i = 0; i < n; ++i) {
        int s = (int)syms[i];
        for (int b = 0; b < len[s]; ++b, ++pos) {
            uint32_t bit = (code[(size_t)s * words + b / 32] >> (31 - b % 32)) & 1u;
            if (bit) out[pos >> 3] |= (uint8_t)(0x80u >> (pos & 7)); /* MSB first :108 */
        }
    }
    return nbytes;
}

/* ------------------------------------------------------------------------- *
 * F. Huffman decode: indices_coding.py:131-168 (remove_padding, decode_text,
 *    decompress_string); F': mask_coding.py:59-96.
 * Greedy prefix match.  Returns symbol count, -1 for an empty file (the
 * reference returns None, :158-159), < -1 on error.
 * ------------------------------------------------------------------------- */
long cgic_oracle_decode(const uint8_t *in, long nbytes, const int32_t *len,
                        const uint32_t *code, int words, int nsym,
                        int64_t *out, long cap)
{
    if (nbytes == 0) return -1;
    int pad = in[0];                                    /* :132-133 */
    long total = (nbytes - 1) * 8;
    long nbits = pad == 0 ? 0 : total - pad;            /* [:-0] is empty in Python */
    if (nbits < 0) nbits = 0;
    /* build a binary trie from the table */
    int maxnodes = 2 * nsym + 2;
    int *child = (int *)malloc(sizeof(int) * 2 * (size_t)maxnodes);
    int *leaf = (int *)malloc(sizeof(int) * (size_t)maxnodes);
    if (!child || !leaf) { free(child); free(leaf); return -2; }
    for (int i = 0; i < maxnodes; ++i) { child[2 * i] = child[2 * i + 1] = -1; leaf[i] = -1; }
    int nn = 1;
    for (int s = 0; s < nsym; ++s) {
        int cur = 0;
        for (int b = 0; b < len[s]; ++b) {
            int bit = (code[(size_t)s * words + b / 32] >> (31 - b % 32)) & 1;
            if (child[2 * cur + bit] < 0) {
                if (nn >= maxnodes) { free(child); free(leaf); return -3; }
                child[2 * cur + bit] = nn++;
            }
            cur = child[2 * cur + bit];
        }
        leaf[cur] = s;
    }
    long cnt = 0;
    int cur = 0;
    for (long p = 0; p < nbits; ++p) {
        int bit = (in[1 + (p >> 3)] >> (7 - (p & 7))) & 1;
        int nx = child[2 * cur + bit];
        if (nx < 0) { /* no code has this prefix: the reference keeps extending
                         current_code forever and emits nothing more */
            break;
        }
        cur = nx;
        if (leaf[cur] >= 0) {
            if (cnt >= cap) { free(child); free(leaf); return -4; }
            out[cnt++] = leaf[cur];
            cur = 0;
        }
    }
    free(child); free(leaf);
    return cnt;
}

/* ------------------------------------------------------------------------- *
 * G. compress() glue for ONE image: CGIC/models/model.py:217-221 (masked select,
 * row-major per granularity), :225-260 (which streams exist per mode),
 * :278-293 (mask -> index scatter, x2 / x4 nearest-upsample sum merge).
 * ------------------------------------------------------------------------- */

/* select: ind [h,w] int64 (fine grid); masks at /4, /2, /1.  Outputs are the
 * three symbol lists; returns counts through n_out[3]. */
void cgic_oracle_select(const int64_t *ind, long h, long w,
                        const int32_t *mc, const int32_t *mm, const int32_t *mf,
                        int64_t *sc, int64_t *sm, int64_t *sf, long *n_out)
{
    long nc = 0, nm = 0, nf = 0;
    for (long y = 0; y < h / 4; ++y)
        for (long x = 0; x < w / 4; ++x)
            if (mc[y * (w / 4) + x] == 1) sc[nc++] = ind[(4 * y) * w + 4 * x];   /* :219 */
    for (long y = 0; y < h / 2; ++y)
        for (long x = 0; x < w / 2; ++x)
            if (mm[y * (w / 2) + x] == 1) sm[nm++] = ind[(2 * y) * w + 2 * x];   /* :220 */
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x)
            if (mf[y * w + x] == 1) sf[nf++] = ind[y * w + x];                    /* :221 */
    n_out[0] = nc; n_out[1] = nm; n_out[2] = nf;
}

/* which of {indices_coarse, indices_medium, indices_fine, mask_coarse,
 * mask_medium} a mode writes (model.py:225-260); bit i set = stream i written */
int cgic_oracle_mode_streams(int mode)
{
    static const int t[7] = { 0x1f, 0x16, 0x0d, 0x0b, 0x01, 0x02, 0x04 };
    return (mode < 0 || mode > 6) ? -1 : t[mode];
}

/*
 * merge (decoder side): decoded symbol lists + decoded masks -> ind [h,w].
 * mc/mm are the decoded mask grids (ignored where the mode does not send
 * them); n* are decoded counts, -1 = the stream file was empty (None).
 * Restates model.py:269-389.  Returns 0, or <0 if a count does not match its
 * mask (the reference raises a shape-mismatch RuntimeError there).
 */
int cgic_oracle_merge(int mode, long h, long w,
                      const int32_t *mc_in, const int32_t *mm_in,
                      const int64_t *sc, long nc, const int64_t *sm, long nm,
                      const int64_t *sf, long nf,
                      int64_t *ind, int32_t *mc_out, int32_t *mm_out, int32_t *mf_out)
{
    long h4 = h / 4, w4 = w / 4, h2 = h / 2, w2 = w / 2;
    /* rebuild the three masks exactly as each mode branch does */
    for (long i = 0; i < h4 * w4; ++i)
        mc_out[i] = (mode == 0 || mode == 2 || mode == 3) ? mc_in[i] : (mode == 4 ? 1 : 0);
    for (long y = 0; y < h2; ++y)
        for (long x = 0; x < w2; ++x) {
            int v;
            if (mode == 0 || mode == 1) v = mm_in[y * w2 + x];
            else if (mode == 3) v = 1 - mc_out[(y / 2) * w4 + x / 2];          /* :332 */
            else v = (mode == 5);
            mm_out[y * w2 + x] = v;
        }
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            int v;
            if (mode == 0) v = 1 - mm_out[(y / 2) * w2 + x / 2] - mc_out[(y / 4) * w4 + x / 4]; /* :280 */
            else if (mode == 1) v = 1 - mm_out[(y / 2) * w2 + x / 2];          /* :302 */
            else if (mode == 2) v = 1 - mc_out[(y / 4) * w4 + x / 4];          /* :320 */
            else v = (mode == 6);
            mf_out[y * w + x] = v;
        }
    /* scatter t[t==1] = decoded, row-major (model.py:287,291,292) */
    int has_c = (mode == 0 || mode == 2 || mode == 3 || mode == 4);
    int has_m = (mode == 0 || mode == 1 || mode == 3 || mode == 5);
    int has_f = (mode == 0 || mode == 1 || mode == 2 || mode == 6);
    long ec = 0, em = 0, ef = 0;
    for (long i = 0; i < h4 * w4; ++i) ec += mc_out[i] == 1;
    for (long i = 0; i < h2 * w2; ++i) em += mm_out[i] == 1;
    for (long i = 0; i < h * w; ++i) ef += mf_out[i] == 1;
    if (has_c && nc >= 0 && nc != ec) return -11;
    if (has_m && nm >= 0 && nm != em) return -12;
    if (has_f && (nf < 0 ? ef != 0 : nf != ef)) return -13;
    long ic = 0, im = 0, jf = 0;
    int64_t *gc = (int64_t *)calloc((size_t)(h4 * w4), sizeof(int64_t));
    int64_t *gm = (int64_t *)calloc((size_t)(h2 * w2), sizeof(int64_t));
    if (has_c && nc >= 0)
        for (long i = 0; i < h4 * w4; ++i) if (mc_out[i] == 1) gc[i] = sc[ic++];
    if (has_m && nm >= 0)
        for (long i = 0; i < h2 * w2; ++i) if (mm_out[i] == 1) gm[i] = sm[im++];
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            int64_t v = 0;
            if (has_f && nf >= 0 && mf_out[y * w + x] == 1) v = sf[jf++];
            v += gm[(y / 2) * w2 + x / 2] + gc[(y / 4) * w4 + x / 4];          /* :293 */
            ind[y * w + x] = v;
        }
    free(gc); free(gm);
    return 0;
}

/* embedding gather -> [C,h,w] (model.py:391-392): exact codebook rows */
int cgic_oracle_gather(const int64_t *ind, long n, const float *cb, int K, int C, float *out)
{
    for (long i = 0; i < n; ++i) {
        if (ind[i] < 0 || ind[i] >= K) return -1;
        for (int c = 0; c < C; ++c) out[c * n + i] = cb[ind[i] * C + c];
    }
    return 0;
}
