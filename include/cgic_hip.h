/*
 * cgic_hip.h -- C ABI of libcgic_hip.so: the MI355X (gfx950) implementation of
 * Control-GIC's granularity-adaptive VQ + router + entropy-coder hot path.
 *
 * The reference (lianqi1008/Control-GIC) is pure Python and has no FFI layer;
 * its boundary for this path is a set of Python classes.  Each entry point
 * below replaces the body of one of those reference methods (cited as
 * file:line relative to the reference root) and is what a binding for that
 * method would call -- see INTEGRATION.md for the ctypes stubs.
 *
 * Conventions
 *  - Plain C: pointers + sizes, no torch / HIP types.  `stream` is a
 *    hipStream_t passed as void* (NULL = the default stream).
 *  - Pointers documented "device" must be device-accessible allocations on
 *    the current HIP device; everything else is host memory.
 *  - All device work is enqueued on `stream`; no entry point synchronises the
 *    host unless its comment says so.  Outputs are written by the kernels on
 *    that stream; inputs are never modified.
 *  - Return value: CGIC_OK (0) or a negative CGIC_ERR_* code; a message for
 *    the calling thread's last error is available from cgic_last_error().
 *  - Tensors are dense row-major ("contiguous") in the layouts the reference
 *    uses (NCHW images/latents, [K,C] codebook).
 */
#ifndef CGIC_HIP_H
#define CGIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGIC_ABI_VERSION 8

#define CGIC_OK 0
#define CGIC_ERR_INVALID (-1)     /* bad argument (shape, ratio, NULL pointer ...) */
#define CGIC_ERR_UNSUPPORTED (-2) /* valid for the reference, not built here (e.g. e_dim != 4) */
#define CGIC_ERR_HIP (-3)         /* a HIP runtime call failed (no device, launch error ...) */
#define CGIC_ERR_NOMEM (-4)
#define CGIC_ERR_CAPACITY (-5)    /* an output / workspace buffer is too small */

typedef void *cgic_stream_t;

/* number of .bin streams compress() can write per image, in this order
 * (CGIC/models/model.py:226-231): indices_coarse, indices_medium,
 * indices_fine, mask_coarse, mask_medium */
#define CGIC_NUM_STREAMS 5

/* Launch resources: kernels whose workgroups hand work to each other (the VQ loss hand-off, the split index streams of
 * grids beyond 64x64 in compress / decompress) use self-resetting ticket slots in library-owned device memory.  Eager
 * launches take them from a ring that belongs to their stream.  Launches being captured into a hipGraph take them from a
 * pool of 262 144 slots per device and keep them while the graph may still be replayed: a VQ launch takes 1 slot, a
 * split-stream compress 6 per image, a split-stream decompress 3 per image (a compress of more than 682 such images falls
 * back to unsplit streams, a decompress of more than 1365 images is cut into several launches).  The caller says when a captured graph is gone:
 *   id = cgic_ticket_scope_begin();  ... capture (on this thread) ...;  cgic_ticket_scope_end();
 *   ... replay for as long as needed ...;  destroy the graph;  cgic_ticket_scope_release(id);     -> slots returned
 * Captures made outside a scope keep their slots for the life of the process (CGIC_ERR_INVALID once the pool is used up).
 * Call each entry point once eagerly on a device before capturing it (allocations and function attributes are set up on
 * first use).  cgic_ticket_scope_begin / _end return the scope id, _release the number of slots returned, _slots_in_use the
 * captured slots currently held on the current device; all return CGIC_ERR_* (< 0) on misuse. */
int cgic_ticket_scope_begin(void);
int cgic_ticket_scope_end(void);
int cgic_ticket_scope_release(int scope);
int cgic_ticket_slots_in_use(void);
/* test tool: synchronises the device and counts the 32-bit words of the current device's ticket memory (every stream's ring + the used
 * part of the pool for captured launches) that are not zero -- 0 whenever nothing is in flight (every user hands its slots back zeroed);
 * < 0: CGIC_ERR_* */
long long cgic_ticket_pool_dirty_words(void);
int cgic_ticket_pool_dirty_dump(unsigned int *out4, int cap);      /* (dev) the first `cap` of them as (where, slot, word, value) */
/* Launch n captured hipGraphs in one call: graph_execs[i] (hipGraphExec_t) on streams[i] (hipStream_t), in order, back to back on the
 * calling thread.  For runtimes of several independent streams of batches (pipeline.LaneStream): from Python every launch is an
 * interpreter round trip and the last lane starts ~100 us after the first. */
int cgic_launch_graphs(void *const *graph_execs, void *const *streams, int n);
/* Launch groups (ABI 6): independent sub-batches of DIFFERENT shapes through ONE launch per kernel.  The reference cuts a high-
 * resolution image into a 768-px grid with a ragged last row / column (inference_high_resolution.py:112-125) and runs
 * model.compress on the tiles one after the other (:236-257); tiles of equal shape are one batch here, but a 2040x1356
 * image is still four shapes = four launch chains.  Between cgic_group_begin(n, shares) and cgic_group_launch(stream) the calls
 * of THIS thread to cgic_entropy_maps_f32 / _u8, cgic_vq_forward_route_f32, cgic_index_histogram, cgic_compress_streams and
 * cgic_decompress_streams check
 * their arguments and record their launches instead of making them (cgic_group_select(g) says which group the following calls belong
 * to; within a group the calls are dependent in call order, across groups nothing is); cgic_group_launch then issues the j-th
 * launches of all groups as ONE launch whose grid is the concatenation of theirs (each workgroup finds its group's argument block
 * from its block index), j = 0, 1, ...  Other entry points, and positions where the groups recorded different kernels, are
 * launched one by one: results never depend on the grouping.  Outputs of recorded calls are written when cgic_group_launch's
 * launches run: nothing else may be enqueued on them in between, and every device buffer handed to a recorded call (inputs,
 * outputs, workspaces) must stay allocated until cgic_group_launch has returned.
 *   shares  host [n] or NULL: each group's share of the chip (sums to ~1; e.g. its share of the latent vectors): the
 *           persistent-workgroup VQ launch of a group takes that share of the CUs.  NULL: 1/n each.
 * cgic_group_launch returns the number of launches issued (>= 0) or CGIC_ERR_*; the group is closed either way.  `stream` must be
 * the stream the recorded calls were given (positions without a grouped form are launched on the stream they recorded).
 * cgic_group_abort closes it without launching.  cgic_group_max: the largest n. */
int cgic_group_max(void);
int cgic_group_begin(int ngroups, const double *shares);
int cgic_group_select(int group);
int cgic_group_launch(cgic_stream_t stream);
void cgic_group_abort(void);
const char *cgic_last_error(void);
int cgic_abi_version(void);
/* number of visible HIP devices, or CGIC_ERR_HIP; never throws, never aborts */
int cgic_device_count(void);

/* ---------------------------------------------------------------------------
 * A. VectorQuantize2.forward -- CGIC/modules/vqvae/quantize.py:69-97
 *
 *   z        device [B, 4, hw]  fp32 (NCHW latent, hw = h*w)
 *   codebook device [K, 4]      fp32 (embedding.weight; K % 16 == 0, K <= 8192)
 *   indices  device [B*hw]      int64   argmin_k ||z - e_k||^2 with the CPU
 *            reference's fp32 rounding sequence; lowest index on exact ties
 *   z_q      device [B, 4, hw]  fp32 or NULL   = z + (e[idx] - z)      (:93)
 *   loss     device [1]         fp32 or NULL   = m + beta*m, m = mean((e[idx]-z)^2)
 *            (:85-90; legacy=1 -> m + beta*m, legacy=0 -> beta*m + m)
 *   hist     device [K]         int64 or NULL  += occurrences of each index
 *            (the usage counter of quantize.py:28,79-81, exact integers)
 *   workspace device, cgic_vq_workspace_bytes(B*hw) bytes, or NULL iff loss==NULL
 *   quant_conv NULL, or the 1x1 convolution in front of the quantiser (CGIC.quant_conv, model.py:51,110) to apply to
 *            every latent vector first: then `z` is the encoder output h and everything above refers to W h + b
 *   prepared NULL, or the image cgic_vq_prepare_f32 made of THIS codebook (16-byte aligned, cgic_vq_prepared_bytes(K)
 *            bytes): what every workgroup otherwise derives from `codebook` at the head of every launch (row norms, the
 *            maxima that fix the fp16 scaling, the split operands of the candidate filter: torch.sum(embedding.weight**2)
 *            of quantize.py:73-75 is recomputed by the reference on every call as well).  Inference keeps one codebook for
 *            thousands of launches: prepare once, pass it along.  The caller re-prepares when the weights change
 *            (VectorQuantize2 keys it on the weight's version counter); results are identical with and without it.
 * Two implementations behind the same contract, bit-identical results: for K % 64 == 0, K <= 1024 (the reference's
 * 1024 x 4 codebook) an fp16-MFMA candidate filter with an exact fp32 resolve, otherwise the fp32-MFMA loop over
 * every code (no fused quant_conv there: CGIC_ERR_UNSUPPORTED).
 * Non-finite values: torch.argmin returns the first NaN; here a NaN distance never wins (lowest index among the
 * minimal non-NaN distances, 0 if all are NaN).  The deviation is confined to vectors whose own distance row
 * contains a non-finite value (tests/test_gpu_stress.py).
 * ------------------------------------------------------------------------- */
/* torch.nn.Conv2d(4, 4, 1) as the CPU reference computes it: per output channel an fma chain over the input
 * channels in order.  oneDNN seeds the accumulator with the bias or adds it at the end depending on shape and
 * thread count (measured with torch 2.10: 1 thread, or a 64x64 latent -> bias last; 8 threads and >= 96x96 ->
 * bias first); `bias_first` selects which of the two sequences to reproduce. */
/* The pixels behind a pair of entropy maps, for the router's threshold-band refinement (cgic_router_f32 below). */
typedef struct cgic_pixels {
    const void *x;      /* device: the image batch the maps were computed from -- [B, 3, 16 h16, 16 w16] fp32, or, with
                         * is_u8, the uint8 frames [B, 16 h16, 16 w16, 3] of cgic_entropy_maps_u8 (4-byte aligned) */
    int is_u8;
    const float *bins;  /* host [32] fp32: torch.linspace(-1, 1, 32) (model.py:480) */
    int nbins;          /* 32 */
    float sigma;        /* 0.01 (model.py:481) */
    const float *flat8; /* device [B, 2 h16, 2 w16] fp32 or NULL: the flat8 output of the entropy call that made the maps.
                         * Without it constant patches are re-evaluated one by one like any other (same result, slower on
                         * flat content) */
    void *scratch;      /* device, cgic_router_refine_scratch_bytes(...) bytes, or NULL (ABI 7).  With it a long band -- smooth or
                         * flat 8-bit content puts tens to hundreds of patches within the band of a threshold -- is not evaluated
                         * by every router workgroup of the image on its own: cgic_router_f32 publishes it to every idle wave of
                         * the launch (the other row bands of a tile, finished router workgroups); cgic_vq_forward_route_f32 has
                         * the row bands of a large tile (>= 32x32 patches routed per image: up to 8 workgroups) split it between
                         * them -- a smooth 768x768 tile 251 -> 82 us (the routers run the plain code first and start over in
                         * the split code only when a band is long: nothing measurable on an ordinary tile).  Same masks either way.
                         * REQUIRED (ABI 8) when the routing segment does not fit the workgroup's LDS (cgic_router_refine_in_lds == 0: the
                         * flattened batch of the reference's encode(), RouterTriple.py:21,40,52,63, or an untiled image beyond
                         * 768x768): the refinement then runs as a chain of launches over patched copies of the maps kept here.
                         * Uninitialised memory; one per launch in flight, and at most FOUR launches that carry one in flight on a
                         * device at a time (the row bands of a tile wait for each other: all of them must be resident) */
    size_t scratch_bytes;
} cgic_pixels;
/* scratch for cgic_pixels.scratch of a router / fused call with these arguments (0: the call takes none; required when
 * cgic_router_refine_in_lds(...) == 0) */
size_t cgic_router_refine_scratch_bytes(int64_t B, int64_t h16, int64_t w16, int per_image);

typedef struct cgic_conv1x1 {
    const float *weight;   /* device [4, 4] fp32 = Conv2d.weight[:, :, 0, 0], row = output channel */
    const float *bias;     /* device [4] fp32 or NULL */
    int bias_first;
} cgic_conv1x1;
/* out[n] = conv(rows[n]) for n rows of 4 floats (post_quant_conv applied to the codebook, model.py:52,115) */
int cgic_conv1x1_rows_f32(const float *rows, int64_t n, const cgic_conv1x1 *conv, float *out, cgic_stream_t stream);

/* Telemetry of the candidate-filter path (K % 64 == 0, K <= 1024): launches enqueued (or captured) after this call add to
 * device_counters[0..3] (uint32, device memory, zeroed by the caller): [0] vectors whose runner-up tile was inside the
 * margin, so that the wave scanned all K codes exactly for them; [1] 64-vector groups with more than 12 such vectors, rerun
 * on the exact fp32-MFMA loop; [2] groups that evaluated a second 16-code candidate set exactly.  NULL switches it off (the
 * default).  Process-wide; the counters are only touched inside those rare branches.  Results never depend on it. */
int cgic_vq_stats(unsigned int *device_counters);
/* Margin telemetry of the candidate filter: the SAME kernel body as cgic_vq_forward_f32's filter path (K % 64 == 0, K <= 1024),
 * instantiated so that it also writes out what it decided on:
 *   scores device [B*hw, K] fp32: the approximate score f_k = ee_k - 2 z.e_k of every code as the fp16 matrix cores delivered it
 *          (scaled back to score units) -- the exactness argument bounds |f_k - F_k| <= 1.8e-6 (ee_k + 2 sum|z_j e_kj|) + floor
 *   aux    device [B*hw, 6]  fp32: per vector (f_min, the margin M, the candidate threshold f_min + M + floor, 1 if the vector was
 *          sent to the all-K exact scan, the exponent q of its score scaling 2^q, the bound S on ee_k + 2 sum|z_j e_kj|)
 *   indices device [B*hw] int64: the result (identical to cgic_vq_forward_f32's)
 * tools/stress_vq.py --telemetry and tests/test_gpu_stress.py compare the observed error and the margin actually consumed by
 * the reference's winners with the budget.  Not a product path: B*hw*K*4 bytes of output. */
int cgic_vq_filter_probe_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K, int64_t *indices,
                             float *scores, float *aux, cgic_stream_t stream);
size_t cgic_vq_workspace_bytes(int64_t n_vectors);
/* bytes of the prepared codebook image (0: this K only has the exact loop, which needs none) / make it (one small launch).
 * Near-duplicate rows -- a trained codebook's clusters and dead codes (quantize.py:22-26,78) -- are packed into one 32-code tile of
 * the image each (keys = original index << 16 | position travel with it; ties still go to the lowest ORIGINAL index), so that they
 * do not send every nearby vector to the all-K exact scan.  For that cgic_vq_prepare_f32 copies the K rows (16 KB) to the host and
 * WAITS for `stream` -- once per codebook; while `stream` is being captured it makes the plain image instead (same results). */
size_t cgic_vq_prepared_bytes(int K);
/* the packing itself, host only (no GPU involved; what cgic_vq_prepare_f32 computes from its host copy of the rows): perm[p] = the
 * ORIGINAL row that sits at position p of the image.  Returns 1 and fills perm[0..K), 0 when there is nothing to pack (no two rows
 * within 3e-4 x the largest |entry| of each other under the max-norm: the image is the plain one), CGIC_ERR_* on bad arguments. */
int cgic_vq_cluster_permutation_host(const float *codebook_host, int K, uint16_t *perm_out);
int cgic_vq_prepare_f32(const float *codebook, int K, int e_dim, void *prepared, cgic_stream_t stream);
int cgic_vq_forward_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K, int e_dim,
                        float beta, int legacy, int64_t *indices, float *z_q, float *loss,
                        int64_t *hist, void *workspace, const cgic_conv1x1 *quant_conv, const void *prepared,
                        cgic_stream_t stream);
/* cgic_vq_forward_f32 and cgic_router_f32 in ONE launch: the router's per-image workgroups share the grid with
 * the VQ workgroups (neither needs the other's output; both need what precedes them, i.e. the latent and
 * the entropy maps).  Same contracts as the two separate calls (`refine`: see cgic_router_f32). */
int cgic_vq_forward_route_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K, int e_dim,
                              float beta, int legacy, int64_t *indices, float *z_q, float *loss,
                              void *workspace, const float *e16, const float *e8, int64_t h16, int64_t w16,
                              double coarse_ratio, double medium_ratio, int per_image, int32_t *mask_c,
                              int32_t *mask_m, int32_t *mask_f, float *gate, int *mode_out,
                              const cgic_conv1x1 *quant_conv, const void *prepared, const cgic_pixels *refine,
                              cgic_stream_t stream);
/* same contract, plain-VALU kernel (no MFMA); kept as an independent
 * implementation for cross-checking the MFMA kernel's rounding */
int cgic_vq_forward_valu_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K,
                             int e_dim, float beta, int legacy, int64_t *indices, float *z_q,
                             float *loss, int64_t *hist, void *workspace, const cgic_conv1x1 *quant_conv,
                             cgic_stream_t stream);

/* Backward of cgic_vq_forward_f32 for training (quantize.py:85-93 under autograd, CGIC.training_step model.py:155-174):
 *   g_zq   device [B, 4, hw] fp32 or NULL   gradient arriving at z_q (straight-through: passes to z)
 *   g_loss device [1] fp32 or NULL          gradient arriving at loss
 *   g_z    device [B, 4, hw] fp32 or NULL   = g_zq + g_loss * (-2/n * w_z) * (e[idx] - z)      (bit-identical to that expression)
 *   g_codebook device [K, 4] fp32 or NULL   = g_loss * (2/n * w_e) * sum_{idx[n] = k} (e[idx] - z_n); n = B*hw*4,
 *          (w_z, w_e) = (1, beta) if legacy else (beta, 1).  Deterministic (fixed-point LDS accumulation per workgroup,
 *          workgroup tables added in a fixed order), unlike torch.index_add_'s fp32 atomics; accurate to ~1e-9 relative
 *          of the largest |e - z| per term.
 *   workspace device, cgic_vq_backward_workspace_bytes(B*hw, K) bytes, or NULL iff g_codebook == NULL */
size_t cgic_vq_backward_workspace_bytes(int64_t n_vectors, int K);
int cgic_vq_backward_f32(const float *z, int64_t B, int64_t hw, const float *codebook, int K, int e_dim,
                         const int64_t *indices, const float *g_zq, const float *g_loss, float beta, int legacy,
                         float *g_z, float *g_codebook, void *workspace, cgic_stream_t stream);

/* usage histogram of an index tensor (quantize.py:79-81): hist[idx[i]] += 1 */
int cgic_index_histogram(const int64_t *indices, int64_t n, int K, int64_t *hist, cgic_stream_t stream);

/* ---------------------------------------------------------------------------
 * C'. Entropy.forward -- CGIC/models/model.py:433-483, patch sizes 8 and 16 in
 * one pass over the image (CGIC.encode computes both, model.py:100-101).
 *
 *   x     device [B, 3, H, W] fp32, H % 16 == 0, W % 16 == 0
 *   bins  host   [32] fp32  (torch.linspace(-1, 1, 32), model.py:480)
 *   e8    device [B, H/8,  W/8 ] fp32 or NULL
 *   e16   device [B, H/16, W/16] fp32 or NULL
 *   flat8 device [B, H/8,  W/8 ] fp32 or NULL: per 8x8 patch its gray value (0.2989 R + 0.5870 G + 0.1140 B, model.py:471) if
 *         all 64 pixels carry the same one, else NaN -- by-product of the pass, consumed by the router's refinement
 *         (cgic_pixels.flat8): constant patches are the large tie groups of real content
 * fp32 throughout; exp / log / reciprocal are the GPU's (v_exp_f32, v_log_f32, v_rcp_f32) and only the 2 bins
 * that bracket a pixel are evaluated (the rest is <= 9e-10 per pixel), so results match the CPU reference to
 * ~1e-6 (measured <= 1.1e-6; tests hold it to 2e-6), not bit-for-bit (SURVEY.md section 7 "hard parts"); see
 * cgic_entropy_maps_ref_f32 below for the variant that follows the reference's rounding.
 * ------------------------------------------------------------------------- */
int cgic_entropy_maps_f32(const float *x, int64_t B, int64_t H, int64_t W, const float *bins,
                          int nbins, float sigma, float *e8, float *e16, float *flat8, cgic_stream_t stream);
/* The same maps in the REFERENCE'S OWN ARITHMETIC (opt-in): torch's CPU operation sequence and summation order -- gray as three
 * products and two sums, IEEE divide by sigma, the mean over a patch's pixels as torch's cascade sum (chunks of 16 pixels in
 * row-major order, then the chunk sums), both sums over the 32 bins as torch's eight strided partials -- with exp / log
 * correctly rounded (fp64, rounded once).  The router's thresholds are k-th smallest entropies with a strict '<'
 * (RouterTriple.py:21-34), so on tie-heavy content (8-bit, flat, blocky images) the last bits of the maps decide which
 * patches fall under a threshold: against the real Entropy class this variant is bit-identical in 98.6-100 % of the values
 * (the rest within 5e-7: torch's exp / log are MKL's) and flips no mask element on any measured content family, where the
 * default kernel (accurate to ~1e-6, order-free) flips a few elements in ~1 of 64 such images.  ~8x the instructions of
 * the default kernel.  Same arguments and contract as cgic_entropy_maps_f32. */
int cgic_entropy_maps_ref_f32(const float *x, int64_t B, int64_t H, int64_t W, const float *bins,
                              int nbins, float sigma, float *e8, float *e16, cgic_stream_t stream);
/* ToTensor + Entropy in one pass, for frames that arrive as uint8 (the datasets of both scripts: PIL image -> center crop ->
 * T.ToTensor(), inference.py:50-59, then CGIC.encode's two Entropy calls, model.py:99-101).
 *   x_hwc  device [B, H, W, 3] uint8 (PIL / decoder layout), 4-byte aligned, H % 16 == 0, W % 16 == 0
 *   x_out  device [B, 3, H, W] fp32 or NULL: byte / 255 exactly as torch's `.div(255)` rounds it = what ToTensor hands to the
 *          conv encoder (bit for bit, all 256 byte values)
 *   e8, e16 as cgic_entropy_maps_f32 -- and bit-identical to cgic_entropy_maps_f32 run on x_out.
 * 3 B read + 12 B written per pixel instead of ToTensor's 3 + 12 and Entropy's 12 again. */
int cgic_entropy_maps_u8(const unsigned char *x_hwc, int64_t B, int64_t H, int64_t W, const float *bins,
                         int nbins, float sigma, float *x_out, float *e8, float *e16, float *flat8, cgic_stream_t stream);

/* ---------------------------------------------------------------------------
 * C. TripleGrainFixedEntropyRouter -- CGIC/modules/vqvae/RouterTriple.py:8-95
 *
 * cgic_router_mode: the 7-way mode from the zero-ness of the ratios, with
 * fine = 1 - coarse - medium evaluated in float64 (:13,19,36,72).
 *
 * cgic_router_f32:
 *   e16 device [B, h16, w16] fp32;  e8 device [B, 2*h16, 2*w16] fp32
 *   per_image = 0: thresholds over the flattened batch (the reference, :21,40,52,63)
 *   per_image = 1: one threshold set per image == B independent B=1 calls
 *   mask_c device [B, h16, w16], mask_m [B, 2h16, 2w16], mask_f [B, 4h16, 4w16] int32
 *   gate   device [B, 4h16, 3*4w16] fp32 or NULL  (cat(up4(gc), up2(gm), gf), :93)
 *   mode_out host int or NULL
 * k = round(n*ratio) uses Python's round-half-even on the float64 product.
 * CGIC_ERR_INVALID if k > n (the reference raises IndexError there).
 *
 *   refine NULL: the masks are exactly the reference router's for the maps AS GIVEN (bit-exact integer / compare work).
 *          Non-NULL (round 4; what the pipeline passes by default): the masks are the reference's from the PIXELS.  The
 *          thresholds are k-th smallest entropies compared with a strict '<' (:21-34) and the maps of cgic_entropy_maps_f32
 *          are within ~1e-6 of the reference's arithmetic, not equal to it, so on tie-heavy content (8-bit, smooth, flat
 *          images) the last bits decide masks and hence .bin bytes.  With the pixels, every patch whose entropy lies within
 *          4e-6 (>= twice the kernel's error) of a threshold is re-evaluated inside the router in the reference's own fp32
 *          operation order (the arithmetic of cgic_entropy_maps_ref_f32), and the k-th smallest is taken again: the result
 *          equals routing on cgic_entropy_maps_ref_f32's maps (tested), at the default kernel's cost whenever the band holds
 *          only the threshold element itself -- the typical image.  A segment whose maps fit the workgroup's LDS
 *          (per-image routing of images / tiles up to 768x768, batch-global routing of a few images:
 *          cgic_router_refine_in_lds) is refined inside the router's workgroup; a larger one (ABI 8: the flattened batch of
 *          the reference's encode(), RouterTriple.py:21,40,52,63; an untiled large image) through patched copies of the
 *          maps in refine->scratch (five launches; not inside a launch group: CGIC_ERR_UNSUPPORTED).  The maps themselves
 *          are not modified.
 * ------------------------------------------------------------------------- */
int cgic_router_mode(double coarse_ratio, double medium_ratio);
/* 1 if cgic_router_f32 / cgic_vq_forward_route_f32 accept `refine` for this shape (ABI 8: every shape whose segment has fewer
 * than 2^31 patches), else 0 */
int cgic_router_refine_supported(int64_t B, int64_t h16, int64_t w16, int per_image);
/* 1 if the segment is refined inside the router's workgroup (its maps fit the LDS); 0: through refine->scratch (see above) */
int cgic_router_refine_in_lds(int64_t B, int64_t h16, int64_t w16, int per_image);
int cgic_router_f32(const float *e16, const float *e8, int64_t B, int64_t h16, int64_t w16,
                    double coarse_ratio, double medium_ratio, int per_image, int32_t *mask_c,
                    int32_t *mask_m, int32_t *mask_f, float *gate, int *mode_out,
                    const cgic_pixels *refine, cgic_stream_t stream);

/* ---------------------------------------------------------------------------
 * D. HuffmanCoding.__init__ / make_heap / merge_nodes / make_codes --
 * CGIC/tools/indices_coding.py:10-17,46-75 (host side; CPython heapq
 * tie-breaking reproduced exactly).
 *
 *   freq  host [n] int64  int(value.item()) per symbol
 *   order host [n] int32 or NULL: order[i] = symbol pushed i-th = iteration
 *         order of the `frequency` mapping (NULL = 0,1,2,...).  NB the
 *         reference's mapping is an nn.ParameterDict built from a plain dict
 *         (quantize.py:28) and therefore iterates its keys sorted AS STRINGS.
 * The table owns host copies and (lazily, per device) device copies.
 * cgic_table_binary(): the fixed table {0:'0', 1:'1'} of BinaryCoding
 * (CGIC/tools/mask_coding.py:11-12).
 * ------------------------------------------------------------------------- */
typedef struct cgic_table cgic_table;
int cgic_table_create(const int64_t *freq, const int32_t *order, int n, cgic_table **out);
int cgic_table_binary(cgic_table **out);
void cgic_table_destroy(cgic_table *t);
int cgic_table_num_symbols(const cgic_table *t);
int cgic_table_max_len(const cgic_table *t);
int cgic_table_words(const cgic_table *t); /* 32-bit words per code = ceil(max_len/32) */
/* host copies: len [n] int32 bits; code [n*words] uint32, bit i of a code is
 * bit 31-(i%32) of word i/32 (MSB first) */
int cgic_table_get(const cgic_table *t, int32_t *len, uint32_t *code);

/* ---------------------------------------------------------------------------
 * E/F. single-stream coders: HuffmanCoding.compress / decompress_string
 * (indices_coding.py:113-126,153-168) and BinaryCoding.compress /
 * decompress_string (mask_coding.py:40-55,81-96), minus the file I/O which
 * stays in the Python host.
 *
 * cgic_encode_stream:
 *   syms   device [n] int64 (a table with 2 symbols also accepts int32 via
 *          elem_bytes = 4: BinaryCoding is fed int32 masks, model.py:230)
 *   out    device [cap] uint8; framing: 1 byte pad count (1..8), code bits
 *          MSB-first, pad zero bits; n == 0 -> 0 bytes (empty file)
 *   nbytes device [1] int32: bytes produced, or CGIC_ERR_* (<0) if a symbol
 *          is outside the table (KeyError in the reference) / cap too small;
 *          `out` must be 4-byte aligned
 * cgic_decode_stream:
 *   in     device [nbytes] uint8 (+ >= 8 readable bytes of slack after it)
 *   syms   device [cap] int64; count device [1] int64: symbols decoded, or -1
 *          for an empty input (the reference returns None)
 * Workspace: cgic_stream_workspace_bytes(n) for encode; decode needs none.
 * ------------------------------------------------------------------------- */
size_t cgic_stream_capacity(const cgic_table *t, int64_t n_symbols);
size_t cgic_stream_workspace_bytes(int64_t n_symbols);
int cgic_encode_stream(const cgic_table *t, const void *syms, int elem_bytes, int64_t n, uint8_t *out,
                       int64_t cap, int32_t *nbytes, void *workspace, cgic_stream_t stream);
int cgic_decode_stream(const cgic_table *t, const uint8_t *in, int64_t nbytes, int64_t *syms,
                       int64_t cap, int64_t *count, cgic_stream_t stream);

/* ---------------------------------------------------------------------------
 * G. CGIC.compress glue, batched -- CGIC/models/model.py:217-260 (encode side)
 * and :269-397 (decode side).  Bit-identical to looping the reference over
 * the batch at B=1.
 *
 * cgic_compress_streams:
 *   ind     device [B, h, w] int64   (VQ indices on the 1/4-resolution grid)
 *   mask_c  device [B, h/4, w/4], mask_m [B, h/2, w/2], mask_f [B, h, w] int32
 *   mode    0..6 (cgic_router_mode)
 *   out     device [B, 5, slot] uint8, slot = cgic_compress_slot_bytes(t, h, w)
 *   nbytes  device [B, 5] int32: length of each stream; -1 = the mode does not
 *           write that stream (model.py:225-260); 0 = written but empty file;
 *           <= -10 = (CGIC_ERR_* - 10) for that stream (a symbol outside the
 *           table -- KeyError in the reference -- or slot too small)
 *   hist    device [n_symbols] int64 or NULL: += occurrences of every index of
 *           ind (all B*h*w of them) -- the usage counter of quantize.py:79-81,
 *           taken in the same launch
 *   Streams are produced by: masked select in row-major order of each
 *   granularity's own grid (ind[:, ::4, ::4][mask_c==1] ..., :219-221),
 *   Huffman coding with `t`, 1-bit packing of mask_c / mask_m (:230-231).
 *
 * cgic_decompress_streams: inverse (model.py:269-397)
 *   in / nbytes as produced above (device)
 *   ind_out device [B, h, w] int64; mask_*_out device int32 (any may be NULL)
 *   codebook device [K, 4] + z_q device [B, 4, h, w] fp32: optional fused
 *           embedding gather (model.py:391-392), exact codebook rows
 *   codebook2 device [K, 4] + z_q2 device [B, 4, h, w] fp32: optional second gather of the same
 *           indices from another table -- post_quant_conv(codebook) gives post_quant_conv(quant)
 *           (CGIC.decode needs both, model.py:114-116) without a pass over the latent
 *   status  device [B] int32: 0, or CGIC_ERR_INVALID when a stream's symbol
 *           count does not match its mask / a mask stream has the wrong length
 *           / an index is outside the codebook (the reference raises there)
 * ------------------------------------------------------------------------- */
size_t cgic_compress_slot_bytes(const cgic_table *t, int64_t h, int64_t w);
size_t cgic_compress_workspace_bytes(int64_t B, int64_t h, int64_t w);
int cgic_mode_streams(int mode); /* bit i set = stream i is written in this mode */
int cgic_compress_streams(const cgic_table *t, const int64_t *ind, const int32_t *mask_c,
                          const int32_t *mask_m, const int32_t *mask_f, int64_t B, int64_t h,
                          int64_t w, int mode, uint8_t *out, int64_t slot, int32_t *nbytes,
                          int64_t *hist, void *workspace, cgic_stream_t stream);
size_t cgic_decompress_workspace_bytes(int64_t B, int64_t h, int64_t w);
/* Which prefix decoder cgic_decompress_streams launches (same results either way; process-wide, read at launch / capture):
 *   CGIC_DECODE_LATENCY     the split-stream decoders: every 64-bit chunk's exit offset for all 64 entry offsets, composed
 *                           across workgroups -- 1-24 workgroups of 1024 threads and 132 KB of LDS per image; shortest time for
 *                           ONE batch on an otherwise idle GPU (B=64 of 256x256: 15 us).  Small launches (every workgroup of the
 *                           decoder and of the merge on a CU of its own: B <= 32 images of 256x256, a few 768x768 tiles) go out
 *                           as ONE launch: the merge workgroups ride behind the decoder's, build their bitsets and prefixes while
 *                           it runs and pick the symbols up through a per-image ticket (B=1: 23.9 -> 20.8 us per call, one
 *                           768x768 tile 36.7 -> 28.6 us).  Workgroups of this mode wait for workgroups of their own launch
 *                           (the decoder's parts for each other, the merge bands for the decoder): meant for ONE stream of
 *                           launches at a time -- several streams decoding concurrently take CGIC_DECODE_THROUGHPUT, whose
 *                           workgroups never wait for another one
 *   CGIC_DECODE_THROUGHPUT  the self-synchronising decoder: one workgroup per image guesses entry offsets and re-walks until
 *                           they agree -- 256 threads and 41 KB of LDS per 256x256 image, 25 us alone, but it leaves the GPU
 *                           to the kernels of other batches in flight (pipeline.LaneStream: 86 -> 101 GPixel/s); the merge
 *                           that follows runs one band of 1024 threads per image instead of four of 512 (half the instructions)
 *   CGIC_DECODE_AUTO        (default) the same kernels as CGIC_DECODE_LATENCY, but the ONE-launch decoder + merge (whose merge bands
 *                           spin on the decoder workgroups of their own launch) only for launches of at most half the chip's
 *                           workgroups: four such launches in flight on four hardware queues are then resident together and
 *                           no band can be left spinning on a decoder that has no CU yet.  CGIC_DECODE_LATENCY is the
 *                           caller's statement that the call has the GPU to itself: it takes the one-launch form up to the
 *                           whole chip
 * Tables with codes longer than 64 bits and grids whose worst case exceeds the LDS budget always take the split-stream
 * / serial paths.  Returns the previous mode, or CGIC_ERR_INVALID. */
#define CGIC_DECODE_AUTO 0
#define CGIC_DECODE_LATENCY 1
#define CGIC_DECODE_THROUGHPUT 2
int cgic_set_decode_mode(int mode);
/* Telemetry of the self-synchronising decoder (CGIC_DECODE_THROUGHPUT): launches enqueued (or captured) after this call add to
 * device_counters[0..2] (uint32, device memory, zeroed by the caller): [0] fix-point sweeps summed over the images, [1] images,
 * [2] the most sweeps one image needed.  NULL switches it off (the default).  Process-wide; results never depend on it. */
int cgic_decode_stats(unsigned int *device_counters);
int cgic_decompress_streams(const cgic_table *t, const uint8_t *in, int64_t slot, const int32_t *nbytes,
                            int64_t B, int64_t h, int64_t w, int mode, int64_t *ind_out,
                            int32_t *mask_c_out, int32_t *mask_m_out, int32_t *mask_f_out,
                            const float *codebook, int K, int e_dim, float *z_q, const float *codebook2,
                            float *z_q2, int32_t *status, void *workspace, int decoder, cgic_stream_t stream);

/* ---------------------------------------------------------------------------
 * Three-grain latent merge in front of the quantiser --
 * CGIC/modules/vqvae/vqvae_blocks.py:361-366:
 *   h = up4(h_coarse)*up4(mask0) + up2(h_medium)*up2(mask1) + h_fine*mask2
 *   h_coarse device [B,C,h/4,w/4], h_medium [B,C,h/2,w/2], h_fine [B,C,h,w] fp32
 *   mask_c/m/f device int32 [B,h/4,w/4] / [B,h/2,w/2] / [B,h,w] (the router's)
 *   out device [B,C,h,w] fp32 -- same products, same left-to-right sums: bit-identical
 * ------------------------------------------------------------------------- */
int cgic_grain_merge_f32(const float *h_coarse, const float *h_medium, const float *h_fine,
                         const int32_t *mask_c, const int32_t *mask_m, const int32_t *mask_f, int64_t B,
                         int C, int64_t h, int64_t w, float *out, cgic_stream_t stream);

/* ---------------------------------------------------------------------------
 * Decoder-side masked blends and the two average pools in front of them --
 * CGIC/modules/vqvae/decoder.py:304-305 (AvgPool2d(4,4,0) / AvgPool2d(2,2,0)), :366-367 (applied to the
 * coarse / medium branch), :372-374 and :375-378 (inside the up path):
 *   medium grid:  h = h * up2(mask0) + h_medium * mask1
 *   fine grid:    h = h * up4(mask0) + h * up2(mask1) + h_fine * mask2
 *   h, h_medium / h_fine, out   device [B,C,hh,ww] fp32 on the grid of the call (out may alias h)
 *   mask_c / mask_m / mask_f    device int32 [B,·,·] at 1/4, 1/2, 1/1 of the FINE grid (the router's / the decoded ones)
 *   same products, same left-to-right sums as the reference expressions: bit-identical.
 * cgic_avgpool_f32: x [planes,H,W] -> out [planes,H/k,W/k], k in {2,4}; row-major running sum of the window
 *   divided by k*k (the order of ATen's CPU kernel: bit-identical to the CPU reference).
 * ------------------------------------------------------------------------- */
int cgic_avgpool_f32(const float *x, int64_t planes, int64_t H, int64_t W, int k, float *out, cgic_stream_t stream);

/* ---------------------------------------------------------------------------
 * Tiling driver: pad + crop of inference_high_resolution.py as ONE pass (ABI 6).  The script pads the image to a multiple of 16
 * with centred zeros (:145-173, :227-228) and crops the padded image tile by tile (:112-125, :236-244); here every tile of every
 * image is written straight from the unpadded image (zeros where a tile reaches into the pad), one launch for all tiles.
 *   x       device fp32 [N,3,H,W], or with is_u8 the uint8 frames [N,H,W,3]
 *   tiles   host [ntiles] (<= 96): dst = device address of the tile of image 0 -- fp32 [3,th,tw] (16-byte aligned) or uint8
 *           [th,tw,3] (4-byte aligned); image_stride = elements from there to the same tile of the next image; (y0, x0) = the
 *           tile's origin in UNPADDED coordinates (negative inside the pad); th, tw with tw % 4 == 0
 * ------------------------------------------------------------------------- */
typedef struct cgic_tile {
    void *dst;
    int64_t image_stride;
    int y0, x0, th, tw;
} cgic_tile;
int cgic_cut_tiles(const void *x, int is_u8, int64_t N, int64_t H, int64_t W, int ntiles, const cgic_tile *tiles,
                   cgic_stream_t stream);
/* cgic_cut_tiles for the tiles of ONE shape + cgic_entropy_maps_f32 / _u8 on them, in one pass: a lane of the map kernel reads its pixels
 * from the source window (zeros inside the pad) and the tile batch is written as a by-product -- 12 B read + 12 B written per pixel
 * instead of 12 + 12 for the cut and 12 again for the maps.  Recorded like the other entropy calls inside a launch group.
 *   src      device fp32 [N,3,H,W] (unpadded), or with is_u8 the uint8 frames [N,H,W,3]
 *   origins  host [T][2] int: (y0, x0) of the T (<= 48) tiles of this shape in unpadded coordinates; the batch is image-major:
 *            tile b of the outputs = (image b / T, tile b % T)
 *   x_out    device fp32 [N*T, 3, th, tw]: the tiles (from frames: byte / 255 as T.ToTensor() rounds it) -- required: it is what the
 *            conv encoder and the router's refinement (cgic_pixels.x, is_u8 = 0) read
 *   e8, e16, flat8 as cgic_entropy_maps_f32 for a batch of N*T images of th x tw; bit-identical to the two separate calls. */
int cgic_entropy_maps_tiles(const void *src, int is_u8, int64_t N, int64_t H, int64_t W, int T, const int *origins,
                            int64_t th, int64_t tw, const float *bins, int nbins, float sigma, float *x_out,
                            float *e8, float *e16, float *flat8, cgic_stream_t stream);
int cgic_decoder_blend_medium_f32(const float *h, const float *h_medium, const int32_t *mask_c, const int32_t *mask_m,
                                  int64_t B, int C, int64_t hh, int64_t ww, float *out, cgic_stream_t stream);
int cgic_decoder_blend_fine_f32(const float *h, const float *h_fine, const int32_t *mask_c, const int32_t *mask_m,
                                const int32_t *mask_f, int64_t B, int C, int64_t hh, int64_t ww, float *out,
                                cgic_stream_t stream);

/* embedding gather on its own (model.py:121,391-392): out[b, c, p] = codebook[ind[b, p], c] */
int cgic_embedding_gather_f32(const int64_t *ind, int64_t B, int64_t hw, const float *codebook, int K,
                              int e_dim, float *out, int32_t *status, cgic_stream_t stream);

/* ---------------------------------------------------------------------------
 * H'. One call per call the reference makes -- CGIC.compress for one image or a batch of images of one size, hot path only
 * (CGIC/models/model.py:206-401 without the conv encoder / decoder; the loop of inference.py:157-166 calls it once per image):
 *   Entropy(8), Entropy(16) on the pixels (model.py:99-101)  ->  [VectorQuantize2.forward + the per-image router, one launch]
 *   (model.py:110-112, vqvae_blocks.py:355)  ->  Huffman / mask coding of the five streams (model.py:217-260)  ->  and, if any
 *   decode output is given, the way back (model.py:269-397): prefix decoder, scatter / merge, codebook gather.
 * Exactly the launches of cgic_entropy_maps_f32 / _u8, cgic_vq_forward_route_f32 (with the threshold-band refinement from these
 * pixels), cgic_compress_streams and cgic_decompress_streams with the same arguments -- bit-identical results -- enqueued by
 * ONE call on `stream`: an eager caller pays one foreign call instead of four (+ their Python around them: B = 1 eager
 * 93 -> the time of the launches themselves).
 *   x        device fp32 [B,3,H,W], or with x_is_u8 the uint8 frames [B,H,W,3]; H, W multiples of 16
 *   z        device fp32 [B,4,H/4,W/4]: the latent behind quant_conv (the conv encoder is the caller's)
 *   e8, e16, flat8   device fp32 [B,H/8,W/8], [B,H/16,W/16], [B,H/8,W/8]: written (the maps are results of the call)
 *   x_out    device fp32 [B,3,H,W] or NULL (x_is_u8 only): ToTensor's output, as cgic_entropy_maps_u8
 *   ind, mask_c/m/f, z_q (or NULL), loss (or NULL), streams [B,5,slot], nbytes [B,5], hist (or NULL): as the single calls
 *   dind ... status: the decode side's outputs (cgic_decompress_streams); dind == NULL: encode only
 *   ws_vq (cgic_vq_workspace_bytes(B*h*w); may be NULL iff loss is), ws_compress (cgic_compress_workspace_bytes),
 *   ws_decompress (cgic_decompress_workspace_bytes; NULL iff dind is), ws_refine (optional: the row bands of large tiles then
 *   split a threshold band between them instead of evaluating it each)
 * ------------------------------------------------------------------------- */
typedef struct cgic_image_io {
    const void *x; int x_is_u8;
    const float *z;
    float *x_out, *e8, *e16, *flat8;
    int64_t *ind; float *z_q, *loss;
    int32_t *mask_c, *mask_m, *mask_f;
    uint8_t *streams; int64_t slot; int32_t *nbytes; int64_t *hist;
    int64_t *dind; int32_t *dmask_c, *dmask_m, *dmask_f; float *dz_q; int32_t *status;
    void *ws_vq, *ws_compress, *ws_decompress;
    void *ws_refine; size_t ws_refine_bytes;     /* cgic_router_refine_scratch_bytes(B, H/16, W/16, 1) bytes or NULL: cgic_pixels.scratch of the call */
} cgic_image_io;
int cgic_compress_image(const cgic_table *t, const float *codebook, int K, int e_dim, const void *prepared, int64_t B, int64_t H,
                        int64_t W, double coarse_ratio, double medium_ratio, float beta, int legacy, const float *bins, int nbins,
                        float sigma, int decoder, const cgic_image_io *io, int *mode_out, cgic_stream_t stream);

/* The same for the tiled driver of high-resolution images (inference_high_resolution.py:112-173 pad + grid, :236-257 the
 * per-tile compress loop): the tiles of N images of one size, grouped by shape (a 2040x1356 image: six tiles of four shapes),
 * through ONE launch chain -- [pad + crop + both entropy maps] -> [VQ + per-tile router] -> stream coder -> (decoder + merge) --
 * every link one launch for all shape groups (launch groups, cgic_group_begin / _launch).  One foreign call per image instead of
 * ~30 recorded ones (eager: 0.54 ms -> the launches themselves).  Per group:
 *   ntiles, th, tw, origins (host [ntiles][2]: (y0, x0) of the tiles in UNPADDED coordinates, may be negative / reach beyond
 *   the image: the centred pad), share (of the chip: the group's pixels / all pixels), and io: the buffers of the group's batch
 *   of N * ntiles tiles (image-major), as cgic_compress_image -- io.x / x_is_u8 are ignored (the tiles come from `src`),
 *   io.x_out (the fp32 tile batch) is REQUIRED, io.z is the latent of the tiles.
 * mode_out: the routing mode (the same for all groups).  At most cgic_group_max() groups. */
typedef struct cgic_tile_group {
    int ntiles, th, tw;
    const int *origins;
    double share;
    cgic_image_io io;
} cgic_tile_group;
int cgic_compress_tiled(const cgic_table *t, const float *codebook, int K, int e_dim, const void *prepared, const void *src,
                        int src_is_u8, int64_t N, int64_t H, int64_t W, int ngroups, const cgic_tile_group *groups,
                        double coarse_ratio, double medium_ratio, float beta, int legacy, const float *bins, int nbins, float sigma,
                        int decoder, int *mode_out, cgic_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CGIC_HIP_H */
