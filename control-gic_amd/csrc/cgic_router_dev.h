// cgic_router_dev.h -- device code of the granularity router (RouterTriple.py:15-95), shared by the
// stand-alone launch (cgic_router.hip) and the horizontally fused VQ+router launch (cgic_vq.hip).
#pragma once
#include "cgic_common.h"
#include "cgic_entropy_dev.h"

#include <math.h>

namespace cgic {

constexpr int kRouterThreads = 1024;   // stand-alone launch; the VQ-fused launch runs the same body with 256

__device__ __forceinline__ uint32_t f2key(float f)
{
    if (f != f) return 0xFFFFFFFFu;                       // NaN sorts last
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k)
{
    if (k == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

// A pointer the compiler KNOWS points into LDS: the maps' copies are reached through pointers that may as well be global (the
// unstaged cases), so plain reads compile to flat_load + a full s_waitcnt per element -- a 768x768 tile's selects spent most of
// their 22 us there.  Reads through this type are ds_read_b32.
typedef const __attribute__((address_space(3))) float *lds_cf32;
__device__ __forceinline__ lds_cf32 as_lds(const float *p) { return (lds_cf32)p; }

struct RouterShared {
    unsigned int hist[4][256];      // pass p counts into [p]; [p + 1] is cleared meanwhile ([0] of the next select during pass 3)
};
constexpr size_t kRouterSharedBytes = sizeof(RouterShared);      // gc_bits follow

// k-th smallest (0-based rank) of n values produced by val(i); all NT threads of the block call.
// 4 passes of 8 bits, ONE barrier per pass and none around them: every wave finds the digit holding the rank by itself from the
// finished histogram (a 256-bin scan is one 16-byte load + one wave scan), so there is nothing to broadcast; pass p counts into
// buffer p while buffer p + 1 is cleared (the last pass clears buffer 0 for the NEXT select: the caller zeroes hist[0] once,
// with a barrier, before the first one -- router_team's staging), and a wave that is still scanning buffer p - 1 disturbs nobody.
// The previous versions: 3 barriers per pass 2.8 + 5.8 us for the two selects of a 256x256 image; a barrier in front and behind.
// what the last pass of a select knew (every wave computes it): the counts of the low key byte among the elements that share
// the threshold's upper 24 key bits, the threshold's own byte, and how many elements with the threshold's key sort before it
struct SelInfo {
    const unsigned int *h;      // [256], valid until pass 2 of the next select clears it
    unsigned int digit, before;
};

// What the first attempt of a two-attempt launch (router_body) hands to the second when it leaves at its COARSE select: both staged maps
// are untouched in LDS, the approximate threshold and the select's last histogram are what the second attempt would compute again
struct RouterResume {
    int at;                     // 0: start from the top; 1: the coarse select's approximate threshold is `thr`, `si` describes its last pass
    float thr;
    SelInfo si;
};

// A lane's pending run of equal digits (see the comment in radix_select's counting loop), for callers that count a select's FIRST pass
// inside a sweep of their own (router_team: the medium map is masked and its exponent byte counted in one go -- COUNTED0 below).
struct RunCount {
    unsigned int run_d = 0, run_n = 0;
    __device__ __forceinline__ void take(unsigned int *h, unsigned int digit)
    {
        if (digit != run_d && run_n) { atomicAdd(&h[run_d], run_n); run_n = 0; }
        run_d = digit;
        ++run_n;
    }
    // pass 0 (the exponent byte): lanes that share the first pending lane's digit hand their counts to it, twice; then what is left
    __device__ __forceinline__ void finish_pass0(unsigned int *h, int lane);
    __device__ __forceinline__ void finish(unsigned int *h) { if (run_n) atomicAdd(&h[run_d], run_n); run_n = 0; }
};

__device__ __forceinline__ void RunCount::finish_pass0(unsigned int *h, int lane)
{
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
        const unsigned long long pend = __ballot(run_n != 0);
        if (!pend) break;
        const int first = __builtin_ctzll(pend);
        const unsigned int d0 = (unsigned int)__builtin_amdgcn_readlane((int)run_d, first);
        const bool same = run_n != 0 && run_d == d0;
        const unsigned int tot = wave_inclusive_scan_u32(same ? run_n : 0u);
        const unsigned int all = (unsigned int)__builtin_amdgcn_readlane((int)tot, 63);
        if (lane == first) atomicAdd(&h[d0], all);
        if (same) run_n = 0;
    }
    finish(h);
}

// COUNTED0: the caller has counted pass 0 (the top key byte of every element) into hist[0], cleared hist[1], and passed a barrier
template <int NT, bool COUNTED0 = false, typename F>
__device__ float radix_select(F val, int64_t n, unsigned int rank0, RouterShared *sh, SelInfo *info = nullptr)
{
    const int tid = threadIdx.x;
    const int lane = lane_id();
    unsigned int prefix = 0, rank = rank0, himask = 0;
    int pass = 0;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8, ++pass) {
        unsigned int *h = sh->hist[pass];
        if (!(COUNTED0 && pass == 0)) {
        // Same-address LDS atomics serialise across the whole workgroup, and entropy values crowd into 2-3 bins of a pass
        // (all of them in the exponent byte): at 9216 values per 768x768 tile a pass spent ~4 us incrementing ONE
        // counter.  Every lane therefore merges equal digits of its own consecutive slots before it touches LDS: a
        // hot pass costs one or two atomics per lane instead of n / NT, a spread-out pass the same as before.
        {
            unsigned int run_d = 0, run_n = 0;
            auto take = [&](float v) {
                const uint32_t key = f2key(v);
                if ((key & himask) != prefix) return;
                const unsigned int digit = (key >> shift) & 0xFF;
                if (digit != run_d && run_n) { atomicAdd(&h[run_d], run_n); run_n = 0; }
                run_d = digit;
                ++run_n;
            };
            // full trips of four elements per lane: all four loads in front of the first atomic (the compiler does not move a
            // load across one; one load per trip was one LDS round trip per element, 9-18 of them per lane and pass for a tile)
            int64_t base = 0;
            for (; base + 4 * NT <= n; base += 4 * NT) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = val(base + u * NT + tid);
#pragma unroll
                for (int u = 0; u < 4; ++u) take(v[u]);
            }
            for (int64_t i = base + tid; i < n; i += NT) take(val(i));
            // The exponent byte (pass 0): nearly every lane of the workgroup is left holding the SAME digit, one atomic each on
            // one counter (~1 cycle apiece, 512-1024 of them per select of a 256x256 image).  The lanes that share the first
            // pending lane's digit hand their counts to it, twice (a masked map has two hot digits: the gated zeros and the rest).
            if (pass == 0) {
#pragma unroll 1
                for (int round = 0; round < 2; ++round) {
                    const unsigned long long pend = __ballot(run_n != 0);
                    if (!pend) break;
                    const int first = __builtin_ctzll(pend);
                    const unsigned int d0 = (unsigned int)__builtin_amdgcn_readlane((int)run_d, first);
                    const bool same = run_n != 0 && run_d == d0;
                    const unsigned int tot = wave_inclusive_scan_u32(same ? run_n : 0u);
                    const unsigned int all = (unsigned int)__builtin_amdgcn_readlane((int)tot, 63);
                    if (lane == first) atomicAdd(&h[d0], all);
                    if (same) run_n = 0;
                }
            }
            if (run_n) atomicAdd(&h[run_d], run_n);
        }
        {
            unsigned int *hz = sh->hist[(pass + 1) & 3];
            for (int i = tid; i < 256; i += NT) hz[i] = 0;
        }
        __syncthreads();
        }
        // every wave: lane handles 4 consecutive digits; find the digit holding `rank`
        // (a router wave of the fused launch only gets the issue slots the VQ workgroup on its CU leaves over: what counts below is
        // the length of the dependent chain -- one 16-byte LDS read, a DPP scan, mask arithmetic, v_readlane -- not the lane count)
        const uint4 c4 = *reinterpret_cast<const uint4 *>(h + 4 * lane);
        const unsigned int p1 = c4.x, p2 = p1 + c4.y, p3 = p2 + c4.z, s = p3 + c4.w;
        const unsigned int incl = wave_inclusive_scan_u32(s);
        const unsigned int excl = incl - s;
        const bool mine = excl <= rank && rank < incl;
        const unsigned int rl = rank - excl;                        // (meaningful in the lane that holds the rank)
        const unsigned int k = (rl >= p1 ? 1u : 0u) + (rl >= p2 ? 1u : 0u) + (rl >= p3 ? 1u : 0u);
        const unsigned int dl = 4u * lane + k, rr = rl - (k == 0 ? 0u : k == 1 ? p1 : k == 2 ? p2 : p3);
        const unsigned long long who = __ballot(mine);
        const int src = who ? __builtin_ctzll(who) : 0;            // (rank >= n cannot happen: checked on the host)
        const unsigned int d = (unsigned int)__builtin_amdgcn_readlane((int)dl, src);
        const unsigned int r = (unsigned int)__builtin_amdgcn_readlane((int)rr, src);
        prefix |= d << shift;
        rank = r;
        himask |= 0xFFu << shift;
        if (shift == 0 && info) { info->h = h; info->digit = d; info->before = r; }
    }
    return key2f(prefix);
}

// the view of the launch's refinement queues (see "evaluating a band's patches with the whole chip" below)
struct RefineQ {
    unsigned int *hdr;          // library-owned slots (kTicketStride words each, acquire_tickets kind 1): slot 2 segment + bank = a select's queue
    unsigned int *board;        // one zero-on-entry ticket slot (kind 0): the launch's board
    unsigned char *scratch;     // caller-owned: per segment {items, values, final index, final value} x {N16, N8} 32-bit words
    unsigned int nq;            // 2 x segments; 0 = no queues (every band is evaluated inside its own workgroup)
    unsigned int pad;
};

struct RouterArgs {
    const float *e16;
    const float *e8;
    int32_t *mask_c, *mask_m, *mask_f;
    float *gate;
    int64_t per;      // images per segment
    int64_t h16, w16;
    int mode;
    unsigned int rank_c;   // 0-based rank of the coarse threshold in the segment
    unsigned int rank_m;
    int stage;             // 1: the segment's e16/e8 are copied to LDS once (all select passes read LDS); 2: only e8 (router_lds_bytes)
    int bands;             // workgroups per segment (per-image segments only): every one finds the thresholds, each writes
                           // only its band of rows of the masks (one CU per 768x768 tile spent 10 us writing 200 KB of masks)
    RefineSrc rf;          // rf.x != nullptr (stage 1 only): threshold-band refinement from the pixels, see refine_select
    RefineQ rq;            // rq.nq != 0: large bands are evaluated by every idle wave of the launch (refine_help_wave)
    unsigned int mg_n8, mg_w8, mg_n4, mg_w4;      // ceil(2^32 / d) for d = n8, w8, n4, w4, or 0: divide (router_prepare)
};

// ---- threshold-band refinement -------------------------------------------------------------------------------------------
// The thresholds are k-th smallest entropies compared with a strict '<' (RouterTriple.py:21-34), and the maps the default
// entropy kernel hands over are within delta ~ 1e-6 of the reference's arithmetic, not equal to it: on tie-heavy content the
// last bits decide masks, hence bytes.  An order statistic is 1-Lipschitz in the sup norm, so with t_a the k-th smallest of
// the approximate values and `band` >= 2 delta: every element below t_a - band is below the true threshold t*, every element
// above t_a + band is above it, and t* is the k-th smallest of the array in which the elements INSIDE the band carry their
// exact values (the others keep their approximations: they stay on their side).  So: count the band; if it holds more than
// the threshold element itself, re-evaluate its patches from the pixels in the reference's own arithmetic
// (cgic_entropy_dev.h), write the values into the LDS copy of the map and select again.  Typical images: band of one, nothing
// to do.  8-bit smooth content: a handful of patches per image.
constexpr float kRefineBand = 4e-6f;       // >= 2 x the default entropy kernel's error (measured <= 1.1e-6, held to 2e-6 by the tests)
// The error bound by VALUE (round 5).  The coarse threshold is the 10 % quantile of the 16x16 entropies: on smooth content it falls
// into the cluster of nearly constant patches, values of 1e-7 .. 1e-4 a few 1e-7 apart, and +-4e-6 around it held up to 41 patches of
// a 256x256 image (334 of a 768x768 tile) -- 60 us of re-evaluation in the image's one router workgroup.  Down there the default
// kernel is much closer to the reference's arithmetic than at large values (what is left is one rounding of the dominant bin's
// p = 1 - eps): measured over every content family, tile and smooth variant of tools/probes/probe_entropy_err.py
//     value   < 1e-5   < 1e-4   < 1e-3   < 1e-2   < 0.1    >= 0.1
//     error   2.8e-7   3.4e-7   4.4e-7   5.8e-7   8.8e-7   1.06e-6
// delta(v) = 7e-7 up to 1e-4, then rising with slope 0.01 to the 2e-6 that holds everywhere (2 x the measured error throughout;
// tests/test_gpu_parity.py holds the kernel to it).  The order-statistic argument goes through with any nondecreasing bound of
// slope < 1: the k elements at or below the approximate threshold t_a are exactly at most t_a + delta(t_a), and an element above
// t_a is exactly at least a - delta(a) >= t_a - delta(t_a), so |t* - t_a| <= delta(t_a); whatever differs from t_a by more than
// delta(t_a) + its own delta keeps its side.  One width for a select: 2 delta(t_a + 4e-6) covers every element within the old band.
__host__ __device__ inline float refine_delta(float v)
{
    const float d = 7e-7f + 0.01f * (v > 1e-4f ? v - 1e-4f : 0.f);
    return d < 2e-6f ? d : 2e-6f;          // (NaN: 2e-6)
}
constexpr int kRefWavesMax = 8;            // waves of a workgroup that evaluate patches (LDS scratch is sized for them): two teams of four
constexpr int kRefListCap = 1024;          // band elements listed per sweep (a longer band is swept range by range)
constexpr int kRefFlatSlots = 128;         // distinct grays of constant patches per select (more: the rest go patch by patch)
constexpr unsigned int kRefEmpty = 0xFFFFFFFFu;      // (a NaN pattern: never the gray of a constant patch)
constexpr int kRefBitWords = 176;          // 64-bit words of the band's member bitmap: 11264 elements, more than any segment whose maps fit kRouterFusedLds

// ---- evaluating a band's patches with the whole chip -------------------------------------------------------------------------
// A band of a few patches is evaluated where it was found (four waves per 16x16 patch, side by side).  A long one -- smooth or flat
// 8-bit content puts tens to hundreds of patches within 4e-6 of a threshold -- used to run 64 pixels per wave and step inside the
// image's ONE router workgroup while the rest of the chip idled (fused launch 24 -> 88 us on smooth 256x256 batches, a smooth
// 768x768 tile 0.56 ms; a unit is ~1.5 us of SIMD time: a CU does 164 units in ~60 us, the chip in one).  Now the owner PUBLISHES
// the patch list, keeps evaluating at its own pace, and waves of the launch that have nothing left to do take patches from it:
// the other row bands of the same tile and the waves of router workgroups that are done.  (The stand-alone router launch only,
// refine_select's RQ: see there for why the fused VQ + router launch keeps the in-workgroup evaluation.)
//
//   queue word (64 bit): (items published << 32) | claims made.  A claim is ONE fetch-and-add of the word (the count of items taken,
//     1 or 4): whoever finds claims < published in the value it gets back owns those items, whatever happened before or after --
//     the word is self-describing, so a late or futile add can only take what is really there or push the claims further past the
//     published count.  (Compare-and-swap on the word as read: with 512 evaluator teams after 10 queues every generation of
//     attempts had one winner -- 5 400 failed claims per launch, 41 patches took 60+ us.)  The owner publishes with a store.
//   payload, in caller-owned scratch: items = (tag << 32 | patch index), results = (tag << 32 | value bits) by patch index -- ONE
//     8-byte write-through store each (sc1: relaxed agent-scope atomics), no counter behind them and nothing to drain: the owner
//     clears its items' result slots before it publishes and polls them afterwards (the tag, a per-queue publication count, tells a
//     result of THIS list from whatever the memory held).  A helper's item costs it four dependent global round trips (queue scan,
//     claim, item, pixels) + the arithmetic; with a done counter behind drained stores it was seven (~17 us per item under load).
//     The per-XCD L2s are not coherent with each other; an agent-scope fence per hand-off would cost ~1.7 us and stall the CU.
//   nobody ever waits for a workgroup that may not have started: the owner waits only for items that were CLAIMED (their
//     claimers are running), a row band that is not the owner (the first to arrive is) waits for that owner, which is running.
//   Header slots come from a pool of their own (acquire_tickets kind 1): zero when first handed out; a finished launch leaves the
//     queue word saying "nothing to claim" (claims >= published), the publication count wherever it got to, and everything else
//     zero again (the last user restores the row-band words).  The board (one zero-on-entry ticket) counts the router workgroups
//     that hold a band open: helpers look for work only while it is non-zero.
enum { QH_CLAIM = 0 /* + 1: the 64-bit queue word */, QH_SEQ = 2, QH_ARRIVE = 3, QH_FIN = 4, QH_LEFT = 5, QH_THR = 6, QH_NFINAL = 7 };
enum { QB_BUSY = 0 };
__host__ __device__ inline size_t refine_scratch_bytes_per_segment(int64_t N16, int64_t N8) { return 24 * (size_t)(N16 + N8); }

__device__ __forceinline__ unsigned int ld_sc1(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float *p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned int *p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float *p, float v) { __hip_atomic_store(reinterpret_cast<unsigned int *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned int add_sc1(unsigned int *p, unsigned int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// per-workgroup LDS of the patch evaluators (router workgroups: inside RefineShared; VQ workgroups: over their codebook image)
struct RefineTeamLds {
    float T[kRefWavesMax][kRefUnitRows * kRefRow];         // chunk sums: the four units of a 16x16 patch are four consecutive waves'
    float rec[kRefWavesMax][kRefRecFloats];
    float P[kRefWavesMax][2 * kBins];
    float bins[kBins];
    unsigned int job_first, job_count;                     // the owner's current round: first item, items
};
struct RefineShared {
    unsigned int cnt[16];                                  // two banks of 8 (coarse / medium select; zeroed once, when the router starts): band size | of
                                                           // which not exactly known | list length | below the band | items published
    unsigned int list[kRefListCap];
    RefineTeamLds tl;
    unsigned int flat_key[kRefFlatSlots];                  // open-addressing set of the constant patches' grays (bit patterns)
    float flat_val[kRefFlatSlots];
    unsigned int small[64];                                // all band elements, when there are at most 64 (the final pick)
    unsigned long long bits[kRefBitWords];                 // MANY: the band's members that are evaluated from their pixels, bit i = element i
    unsigned int pre[kRefBitWords];                        //       members in front of each word
    float thr;
    unsigned int busy_held;                                // this workgroup has raised the board's busy count
    unsigned int tag;                                      // the current list's publication count
    unsigned int flag;
};

__device__ __forceinline__ unsigned int flat_hash(unsigned int key) { return (key * 2654435761u) >> 25; }      // 7 bits
// slot of `key` in the set, or -1
__device__ __forceinline__ int flat_find(const unsigned int *keys, unsigned int key)
{
    unsigned int s = flat_hash(key);
    for (int t = 0; t < kRefFlatSlots; ++t, s = (s + 1) & (kRefFlatSlots - 1)) {
        const unsigned int k = keys[s];
        if (k == key) return (int)s;
        if (k == kRefEmpty) return -1;
    }
    return -1;
}
// the slot `key` sits in afterwards, or -1 (full: the element is evaluated patch by patch)
__device__ __forceinline__ int flat_insert(unsigned int *keys, unsigned int key)
{
    unsigned int s = flat_hash(key);
    for (int t = 0; t < kRefFlatSlots; ++t, s = (s + 1) & (kRefFlatSlots - 1)) {
        const unsigned int old = atomicCAS(&keys[s], kRefEmpty, key);
        if (old == kRefEmpty || old == key) return (int)s;
    }
    return -1;
}

// one unit: lane = pixel `lane` of quarter q of patch e of the segment whose first image is img0 (an 8x8 patch is its own single
// quarter) -> chunk sums in T.  wP, nP: patches per image row / per image.
template <int P>
__device__ __forceinline__ void refine_unit(const RefineSrc &rf, const float *bins, int64_t img0, int wP, int nP, int e, int q, float *rec, float *T)
{
    const int lane = lane_id();
    const int bi = e / nP, r = e - bi * nP;
    const int py = r / wP, px = r - py * wP;
    int64_t y, x;
    if (P == 16) { y = 16 * (int64_t)py + 4 * q + (lane >> 4); x = 16 * (int64_t)px + (lane & 15); }
    else         { y = 8 * (int64_t)py + (lane >> 3);          x = 8 * (int64_t)px + (lane & 7); }
    const float gray = ref_gray_at(rf, img0 + bi, y, x);
    int j0;
    float v[kRefWin];
    ref_pixel(bins, rf.sigma, gray, j0, v);
    ref_unit_chunks(rec, j0, v, T);
}

// what a helper needs to know about the launch
struct RoundCtx {
    RefineQ rq;
    RefineSrc rf;
    int64_t per;
    int h16, w16;
};
__device__ __forceinline__ RoundCtx round_ctx(const RouterArgs &a)
{
    RoundCtx c;
    c.rq = a.rq; c.rf = a.rf; c.per = a.per; c.h16 = (int)a.h16; c.w16 = (int)a.w16;
    return c;
}
// where a select's queue lives
struct RefineQView {
    unsigned int *hdr;
    unsigned long long *items;   // [published] (tag << 32 | patch index of the segment)
    unsigned long long *vals;    // [n] result by patch index: (tag << 32 | value bits)
    unsigned int *fidx;          // [<= n] the final (index, value) list a tile's row bands share
    float *fval;
};
__device__ __forceinline__ RefineQView refine_qview(const RefineQ &rq, int64_t N16, int q)
{
    const int64_t N8 = 4 * N16;
    const int seg = q >> 1, bank = q & 1;
    const size_t n = (size_t)(bank ? N8 : N16);
    unsigned char *base = rq.scratch + (size_t)seg * refine_scratch_bytes_per_segment(N16, N8) + (bank ? 24 * (size_t)N16 : 0);
    RefineQView v;
    v.hdr = rq.hdr + (size_t)q * kTicketStride;
    v.items = reinterpret_cast<unsigned long long *>(base);
    v.vals = reinterpret_cast<unsigned long long *>(base + 8 * n);
    v.fidx = reinterpret_cast<unsigned int *>(base + 16 * n);
    v.fval = reinterpret_cast<float *>(base + 20 * n);
    return v;
}
__device__ __forceinline__ unsigned long long *refine_qword(const RefineQ &rq, int q)
{
    return reinterpret_cast<unsigned long long *>(rq.hdr + (size_t)q * kTicketStride + QH_CLAIM);
}
__device__ __forceinline__ unsigned long long granule(unsigned int tag, float v) { return ((unsigned long long)tag << 32) | __float_as_uint(v); }

#ifdef CGIC_PHASE_CLOCKS      // dev: g_phase_clk[20..]: sum of helper item times [10 ns] | helper claims that found nothing | - | items helpers took | VQ waves that helped | publications
#define CGIC_RQ_COUNT(k, n) atomicAdd((unsigned long long *)&g_phase_clk[20 + (k)], (unsigned long long)(n))
// dev: per (segment, select) of the first 64 segments: [0] published [1] own rounds exhausted [2] all done [3] items published [4] of which this workgroup's own
#define CGIC_RQ_STAMP(qid, k, v) do { if (threadIdx.x == 0 && (qid) < 128) g_blk_t[4096 + 8 * (qid) + (k)] = (v); } while (0)
// dev: the MANY path of a select without queues, row band 0 (tools/probes/probe_tile_timeline.py): the same slots
#define CGIC_RS_STAMP(k) do { if (!RQ && threadIdx.x == 0 && band == 0 && qid < 128) g_blk_t[4096 + 8 * qid + (k)] = wall_clock64(); } while (0)
// dev: the front of a select's refinement, row band 0: [0] entry [1] past the histogram shortcut [2] counted [3] past the early returns
#define CGIC_RE_STAMP(k) do { if (!RQ && threadIdx.x == 0 && band == 0 && qid < 128) g_blk_t[4096 + 1024 + 8 * qid + (k)] = wall_clock64(); } while (0)
#else
#define CGIC_RE_STAMP(k) do {} while (0)
#define CGIC_RS_STAMP(k) do {} while (0)
#define CGIC_RQ_COUNT(k, n) do {} while (0)
#define CGIC_RQ_STAMP(qid, k, v) do {} while (0)
#endif

// One wave with nothing left to do looks for a queue with unclaimed items -- own_q first (-1: none), then the launch's queues,
// lane = queue, one drawn with probability proportional to what it has left (so that the helpers spread like the work does) --
// takes one 16x16 patch or up to four 8x8 ones, evaluates them (64 pixels per step) and stores the results.  Returns whether it
// found anything.  T, rec, P: this wave's LDS scratch; bins: the bin centres in LDS.  No barrier inside.
static __device__ __forceinline__ bool refine_help_wave(const RoundCtx &c, float *T, float *rec, float *P, const float *bins, int own_q,
                                                        unsigned int salt)
{
    const int lane = lane_id();
    const int nq = (int)c.rq.nq;
#ifdef CGIC_PHASE_CLOCKS
    const long long dbg_t0 = wall_clock64();
#endif
#pragma unroll 1
    for (int attempt = 0; attempt < 4; ++attempt) {
        int cand = -1;
        if (own_q >= 0) {
            const unsigned long long w = ld_sc1(refine_qword(c.rq, own_q));
            if ((unsigned int)w < (unsigned int)(w >> 32)) cand = own_q;
        }
        if (cand < 0) {
            const unsigned int rnd = ((unsigned int)clock64() ^ (salt * 2654435761u)) + (unsigned int)(attempt * 97);
            const int nchunk = (nq + 63) >> 6;
            const int c0 = (int)(salt % (unsigned int)nchunk);
#pragma unroll 1
            for (int ci = 0; ci < nchunk && cand < 0; ci += 2) {       // (two chunks per round trip)
                int ch0 = c0 + ci, ch1 = c0 + ci + 1;
                if (ch0 >= nchunk) ch0 -= nchunk;
                if (ch1 >= nchunk) ch1 -= nchunk;
                const int q0 = 64 * ch0 + lane, q1 = 64 * ch1 + lane;
                unsigned long long w0 = 0, w1 = 0;
                if (q0 < nq) w0 = ld_sc1(refine_qword(c.rq, q0));
                if (ci + 1 < nchunk && q1 < nq) w1 = ld_sc1(refine_qword(c.rq, q1));
#pragma unroll
                for (int h = 0; h < 2 && cand < 0; ++h) {
                    const unsigned long long w = h ? w1 : w0;
                    const unsigned int wl = (unsigned int)w, wh = (unsigned int)(w >> 32);
                    const unsigned int left = wl < wh ? wh - wl : 0u;
                    const unsigned int incl = wave_inclusive_scan_u32(left);
                    const unsigned int total = (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
                    if (total) {
                        const unsigned int r = rnd % total;
                        const unsigned long long hit = __ballot(left != 0 && incl - left <= r && r < incl);
                        const int src = hit ? __builtin_ctzll(hit) : 0;
                        cand = __builtin_amdgcn_readlane(h ? q1 : q0, src);
                    }
                }
            }
        }
        if (cand < 0) return false;
        const unsigned int take = (cand & 1) ? 4u : 1u;      // (the word as it comes back says how many of them are really there)
        unsigned int lo = 0, hi = 0;
        if (lane == 0) {
            const unsigned long long old = __hip_atomic_fetch_add(refine_qword(c.rq, cand), (unsigned long long)take, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT);
            lo = (unsigned int)old; hi = (unsigned int)(old >> 32);
        }
        lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)lo);
        hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)hi);
        if (lo >= hi) { if (lane == 0) CGIC_RQ_COUNT(1, 1); continue; }
        const unsigned int got = hi - lo < take ? hi - lo : take;
        const int n16 = c.h16 * c.w16;
        const RefineQView v = refine_qview(c.rq, c.per * n16, cand);
        const int64_t img0 = (int64_t)(cand >> 1) * c.per;
        unsigned long long it[4];
#pragma unroll
        for (unsigned int k = 0; k < 4; ++k) it[k] = k < got ? ld_sc1(v.items + lo + k) : 0ull;      // (all loads in front of the first use)
#pragma unroll 1
        for (unsigned int k = 0; k < got; ++k) {
            const unsigned long long item = k == 0 ? it[0] : k == 1 ? it[1] : k == 2 ? it[2] : it[3];
            const int e = (int)(unsigned int)item;
            const unsigned int tag = (unsigned int)(item >> 32);
            float ent;
            if (cand & 1) {
                refine_unit<8>(c.rf, bins, img0, 2 * c.w16, 4 * n16, e, 0, rec, T);
                ent = ref_finalize(ref_add_rows(0.f, T), 64, P);
            } else {
                float acc = 0.f;
                for (int q = 0; q < 4; ++q) {
                    refine_unit<16>(c.rf, bins, img0, c.w16, n16, e, q, rec, T);
                    acc = ref_add_rows(acc, T);
                }
                ent = ref_finalize(acc, 256, P);
            }
            if (lane == 0) st_sc1(v.vals + e, granule(tag, ent));
        }
#ifdef CGIC_PHASE_CLOCKS
        if (lane == 0) { CGIC_RQ_COUNT(0, wall_clock64() - dbg_t0); CGIC_RQ_COUNT(3, got); }
#endif
        return true;
    }
    return false;
}

// The waves of a router workgroup that is done evaluate other images' band patches while a router holds a band open (stand-alone
// launch).  `tl`: the workgroup's evaluator LDS (bins are set).  Waves beyond the LDS scratch leave at once.
// max_waves: how many waves of this workgroup look for work (every claim is a fetch-and-add on the queue's one word: the eight waves of
// each of a launch's ~60 finished routers, all after the same few dozen items, spent more time in line at that word than evaluating)
__device__ __forceinline__ void refine_help_while_busy(const RouterArgs &a, RefineTeamLds *tl, int max_waves = kRefWavesMax)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave >= kRefWavesMax || wave >= max_waves) return;
    const RoundCtx c = round_ctx(a);
#pragma unroll 1
    for (;;) {
        if (refine_help_wave(c, tl->T[wave], tl->rec[wave], tl->P[wave], tl->bins, -1, blockIdx.x * 16u + (unsigned int)wave)) continue;
        if (ld_sc1(a.rq.board + QB_BUSY) == 0) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

// is_exact of refine_select: element i of the masked medium map is a gated zero -- exactly known, never re-evaluated -- when its
// coarse parent's gate bit is set
struct ExactGate {
    const unsigned long long *gc_bits;      // nullptr: nothing is exact
    int n8i, w8i, n16i, w16i;
    unsigned int mg_n8, mg_w8;
    __device__ __forceinline__ bool operator()(int i) const
    {
        if (!gc_bits) return false;
        const int b = mg_n8 ? (int)__umulhi((unsigned int)i, mg_n8) : i / n8i, r = i - b * n8i;
        const int y = mg_w8 ? (int)__umulhi((unsigned int)r, mg_w8) : r / w8i, x = r - y * w8i;
        const int c = b * n16i + (y >> 1) * w16i + (x >> 1);
        return (gc_bits[c >> 6] >> (c & 63)) & 1ull;
    }
};

// One select's refinement.  arr: the LDS copy of the map (n values; element i = patch i of the segment's images in order),
// t_a: its rank-th smallest value as found, is_exact(i): the value is exact already (a gated zero of the masked medium map).
// Returns the threshold to use; arr is patched in place.  All NT threads call; contains barriers.
//   1. count the band (and note its members while they are few); the threshold element alone -> done.
//   FEW (<= 64 members, the usual case): evaluate the members, then pick the (rank - below)-th smallest among them with one wave.
//   MANY (flat / smooth content): constant patches first (flat8 says so without touching a pixel): one evaluation per DISTINCT
//        gray (its first member stands for all); the others range by range; then a full select over the patched copy.
// A patch is 64 pixels per wave and step: an 8x8 patch is one step, a 16x16 one four (quarters of four rows).  A list that this
// workgroup's waves finish in two steps is evaluated here (four waves per 16x16 patch, side by side); a longer one is published
// to the launch's queue (see above) when there is one, else swept by one wave per patch.
// Row bands (nb > 1 workgroups per segment, each repeating the selects): with a queue, the first band to arrive at a refinement
// that goes to the queue owns it; the others evaluate what it publishes and take over its final (index, value) list and threshold.
// RQ: with the launch's refinement queues (the stand-alone router launch).  The fused VQ + router launch instantiates RQ = false: the
// queue code inlined into its router workgroups cost the router's ORDINARY path -- that launch's critical path -- 3 to 12 us
// through register allocation (B = 64 x 256x256: 23.5 -> 29.4 us with the owner's side inlined, 35.8 with the row bands' side as
// well, no refinement even running), a real function call is not an option (the callee's register count becomes the kernel's:
// 248, one wave per SIMD), and the VQ workgroups' last waves as helpers bought nothing in a stream of batches (the long launch
// overlaps the other lanes' work either way, and helpers hold CUs those lanes want: 68 vs 82 GPixel/s on smooth 8-bit batches).
// BAIL (attempt one of a launch whose row bands can split a band, see router_body): a band with more than four rounds of this
// workgroup's waves to evaluate from pixels is not done here -- *bail_out = true, and the caller starts over in the SPLIT instantiation.
template <int NT, int P, bool RQ, bool SPLIT, bool BAIL = false>
__device__ __forceinline__ float refine_select(const RouterArgs &a, int qid, int nb, int band, float *arr, int n, unsigned int rank, float t_a,
                                                       ExactGate is_exact, int64_t img0, int wP, int nP, RefineShared *rs_, RouterShared *sh,
                                                       const SelInfo &si, bool *bail_out = nullptr, bool *refined_out = nullptr, bool precounted = false)
{
    RefineShared *rs = rs_;
    const RefineSrc &rf = a.rf;
    const int bank = qid & 1;
    unsigned int *cnt = rs->cnt + 8 * bank;
    constexpr int NW = NT / 64, NWR = NW < kRefWavesMax ? NW : kRefWavesMax;
    constexpr int UPP = P == 16 ? 4 : 1;              // units (64 pixels, one wave and step) per patch
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const float w = 2.f * refine_delta(t_a + kRefineBand);
    CGIC_RE_STAMP(0);
    // (precounted: the first attempt of a two-attempt launch left at this select AFTER the count below -- the counters and the list of
    // the first 64 members are as it left them, router_team did not reset them)
    if (!precounted) {
    // 0. The usual image leaves here WITHOUT a pass over the map and without a barrier: the select's last pass counted the low
    // key byte of everything that shares the threshold's upper 24 key bits, and a band of +-4e-6 around a value in [0.13, 4) is
    // at most +-250 such steps -- when both band edges share those 24 bits with the threshold (most of the time for values
    // above 1), the band's size and the number of its members sorting before the threshold element are sums over that
    // histogram, which every wave still has in LDS.  (Measured on the timed batch: the counting pass below cost the router's
    // critical path +1.6 us per select whether or not anything was in the band.)
    {
        const uint32_t kt = f2key(t_a), klo = f2key(t_a - w), khi = f2key(t_a + w);
        if (t_a == t_a && ((klo ^ kt) >> 8) == 0 && ((khi ^ kt) >> 8) == 0) {          // (wave-uniform: t_a is)
            const unsigned int dlo = klo & 255u, dhi = khi & 255u, dt = si.digit;
            unsigned int mall = 0, before = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int d = 4u * lane + j, c = si.h[d];
                mall += (d >= dlo && d <= dhi) ? c : 0u;
                before += (d >= dlo && d < dt) ? c : 0u;
            }
            // both sums in one DPP scan (a segment that is refined has < 2^16 elements: it fits the fused launch's LDS budget):
            // twelve dependent ds_bpermute round trips were ~0.4 us of every select's tail
            const unsigned int both = (unsigned int)__builtin_amdgcn_readlane((int)wave_inclusive_scan_u32((mall << 16) | before), 63);
            mall = both >> 16;
            before = (both & 0xFFFFu) + si.before;     // + the elements EQUAL to the threshold that sort before it
            if (__builtin_expect(mall < 2u || before == 0u, 1)) return t_a;      // the threshold element alone / nothing of the band below it (see 1.)
        }
    }
    CGIC_RE_STAMP(1);
    {       // (cnt and the bin centres were set up when the router started: no barrier in front of the count)
        unsigned int minx = 0, below = 0;
        for (int i = tid; i < n; i += NT) {
            const float d = arr[i] - t_a;
            const bool in = fabsf(d) <= w;                         // (NaN: false)
            if constexpr (SPLIT || RQ) {
                // (one atomic per wave: the 6408 constant patches of a flat tile's band were 8.7 us on this one counter)
                const unsigned long long m = __ballot(in);
                if (m) {
                    const int first = __builtin_ctzll(m);
                    unsigned int base = 0;
                    if (lane == first) base = atomicAdd(&cnt[0], (unsigned int)__builtin_popcountll(m));
                    base = (unsigned int)__shfl((int)base, first, kWave);
                    const unsigned int s = base + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (in) {
                        if (s < 64) rs->small[s] = (unsigned int)i;
                        minx += is_exact(i) ? 0u : 1u;
                    }
                }
            } else if (in) {
                const unsigned int s = atomicAdd(&cnt[0], 1u);
                if (s < 64) rs->small[s] = (unsigned int)i;
                const bool ex = is_exact(i);
                minx += ex ? 0u : 1u;
                if constexpr (BAIL) {       // cnt[7]: members that are evaluated from their pixels (not exact, not a constant patch)
                    if (!ex) {
                        bool constant = false;
                        if (const float *fl = a.rf.flat8) {
                            fl += img0 * (P == 16 ? 4 * (int64_t)nP : (int64_t)nP);
                            if (P == 8) constant = fl[i] == fl[i];
                            else {
                                const int bi = i / nP, r = i - bi * nP, py = r / wP, px = r - py * wP;
                                const float *q = fl + (int64_t)bi * 4 * nP + (int64_t)(2 * py) * (2 * wP) + 2 * px;
                                constant = q[0] == q[1] && q[0] == q[2 * wP] && q[0] == q[2 * wP + 1];
                            }
                        }
                        if (!constant) atomicAdd(&cnt[7], 1u);
                    }
                }
            }
            below += d < -w ? 1u : 0u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d, kWave);
        if (lane == 0 && below) atomicAdd(&cnt[3], below);
        if (minx) atomicAdd(&cnt[1], minx);
    }
    }
    __syncthreads();
    CGIC_RE_STAMP(2);
    const unsigned int m_all = cnt[0], m_inexact = cnt[1], c_below = cnt[3];
    if (__builtin_expect(m_all < 2 || m_inexact == 0, 1)) return t_a;          // (workgroup-uniform) the threshold element alone: nothing can change
    // The band's members all rank at or above the threshold element (nothing of the band sorts before it): the true threshold is
    // the SMALLEST exact value of the band, and under the strict '<' no member of the band lies below that -- every one of them
    // gets 0 whatever the exact values are, everything outside the band keeps its side: t_a already gives the reference's mask.
    // (Half of the two-member bands of noise-like content.)
    if (__builtin_expect(rank == c_below, 1)) return t_a;
    // From here on this workgroup is the launch's critical path (in the fused launch it shares its CU with an issue-bound VQ
    // workgroup): its instructions go first.  Measured, 64 images of 256x256, fused launch: routers at priority 3 throughout
    // 25.4 -> 27.4 us when no image refines (the VQ workgroups pay), refining images 35.5 -> 30.3 us; raised only here: both.
    __builtin_amdgcn_s_setprio(3);
    CGIC_RE_STAMP(3);
    if (refined_out) *refined_out = true;         // (this select's band is evaluated from pixels: the image is not an ordinary one)

    const bool have_q = RQ && a.rq.nq != 0;
    // Without the queues (the fused launch), the row bands of a tile -- nb workgroups that all find the same band -- at least
    // SPLIT it: every band evaluates its share of the members (FEW: by a hash of the patch index; MANY: every nb-th in index
    // order) and publishes the values (by patch index, write-through, drained, then its count on the select's header); every
    // band waits for the count to reach the total and reads the others' values.  (They used to evaluate every member each: a
    // smooth 768x768 tile 251 us of the fused launch.)
    // The bands of a tile are all in the launch's grid in front of (or beside) VQ workgroups that wait for nobody: each gets a CU.
    const bool split = SPLIT && !RQ && nb > 1 && a.rq.nq != 0;
    RefineQView qv;
    if (have_q || split) qv = refine_qview(a.rq, a.per * a.h16 * a.w16, qid);
    float *xvals = reinterpret_cast<float *>(qv.vals);          // split: values by patch index
    // (ownership by a multiplicative hash of the patch index: balanced for any nb, no division)
    auto mine_of = [&](int e) { return !split || (((((unsigned int)e * 0x9E3779B1u) >> 16) * (unsigned int)nb) >> 16) == (unsigned int)band; };
    // split: `total` members are evaluated by all bands together, `own` of them here -> wait for the rest, then leave the header clean
    auto exchange = [&](unsigned int total, unsigned int own) {
        drain_stores();
        __syncthreads();
        if (tid == 0) {
            if (own) add_sc1(qv.hdr + QH_ARRIVE, own);
            while (ld_sc1(qv.hdr + QH_ARRIVE) < total) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    };
    auto exchange_leave = [&]() {
        __syncthreads();
        if (tid == 0 && add_sc1(qv.hdr + QH_LEFT, 1u) == (unsigned int)nb - 1u) {      // the last band out restores the header
            st_sc1(qv.hdr + QH_ARRIVE, 0u);
            st_sc1(qv.hdr + QH_LEFT, 0u);
        }
    };
    const RoundCtx rc = round_ctx(a);
    bool share = false;           // this workgroup owns a refinement whose outcome the segment's other row bands take over

    auto unit = [&](int e, int q, float *T) { refine_unit<P>(rf, rs->tl.bins, img0, wP, nP, e, q, rs->tl.rec[wave], T); };
    // Evaluate the patches lst[0..L) that `todo(k, e)` accepts (Lw of them) and hand each result to out(k, e, value).
    auto sweep = [&](const unsigned int *lst, int L, int Lw, auto todo, auto out) {
        if (RQ && have_q && Lw * UPP > 4 * NWR) {
            // ---- through the launch's queue.  The list in queue order, in LDS for this workgroup and in the scratch for everybody else
            unsigned int *qitems = (Lw == L && lst == rs->list) ? rs->list : rs->list + kRefListCap / 2;
            if (tid == 0) rs->tag = ld_sc1(qv.hdr + QH_SEQ) + 1u;         // (only a queue's owner touches its publication count)
            __syncthreads();
            const unsigned int tag = rs->tag;
            if (qitems == lst) {
                for (int k = tid; k < L; k += NT) {
                    const unsigned int e = lst[k];
                    st_sc1(qv.vals + e, 0ull);                             // (tags start at 1: whatever the slot held, it is not of this list)
                    st_sc1(qv.items + k, ((unsigned long long)tag << 32) | e);
                }
            } else {
                for (int k = tid; k < L; k += NT) {
                    const unsigned int e = lst[k];
                    if (todo(k, (int)e)) {
                        const unsigned int pos = atomicAdd(&cnt[4], 1u);
                        qitems[pos] = e;
                        st_sc1(qv.vals + e, 0ull);
                        st_sc1(qv.items + pos, ((unsigned long long)tag << 32) | e);
                    }
                }
            }
            drain_stores();
            __syncthreads();
            const unsigned int Lq = (unsigned int)Lw;
            // The owner takes its own items a round at a time -- NWR units: two 16x16 patches (four waves each, side by side) or
            // eight 8x8 ones -- at the pace of the in-workgroup sweep: the items are in LDS, and the ONE fetch-and-add per round
            // that keeps the launch's idle waves off them is issued a round ahead.  It never sits on more than two rounds:
            // whatever it has not claimed those waves take as soon as they are free (refine_help_wave).
            constexpr unsigned int PER = (unsigned int)(NWR / UPP);       // patches per round
            auto claim = [&]() -> unsigned long long {
                return __hip_atomic_fetch_add(refine_qword(a.rq, qid), (unsigned long long)PER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            auto post = [&](unsigned long long old) {        // thread 0: what the claim got -> the round's slot
                const unsigned int lo = (unsigned int)old, hi = (unsigned int)(old >> 32);
                rs->tl.job_first = lo;
                rs->tl.job_count = lo < hi ? (hi - lo < PER ? hi - lo : PER) : 0u;
            };
            if (tid == 0) {
                if (!rs->busy_held) { rs->busy_held = 1; add_sc1(a.rq.board + QB_BUSY, 1u); }
                st_sc1(qv.hdr + QH_SEQ, tag);
                st_sc1(refine_qword(a.rq, qid), (unsigned long long)Lq << 32);
                CGIC_RQ_COUNT(5, 1);
                CGIC_RQ_STAMP(qid, 0, wall_clock64());
                CGIC_RQ_STAMP(qid, 3, Lq);
                post(claim());
            }
            __syncthreads();
            unsigned int mine = 0;
#pragma unroll 1
            for (;;) {
                const unsigned int got = rs->tl.job_count, first = rs->tl.job_first;
                __syncthreads();
                if (!got) break;
                unsigned long long nextc = 0;
                if (tid == 0) nextc = claim();                  // (the next round: its round trip runs under this one's evaluation)
                const int slot = wave / UPP;                    // which of the round's patches this wave works on
                const bool go = wave < NWR && (unsigned int)slot < got;
                const int e = go ? (int)qitems[first + slot] : 0;
                if (go) unit(e, wave % UPP, rs->tl.T[wave]);
                if (UPP > 1) __syncthreads();
                if (go && wave % UPP == 0) {
                    float acc = 0.f;
                    for (int q = 0; q < UPP; ++q) acc = ref_add_rows(acc, rs->tl.T[wave + q]);
                    const float ent = ref_finalize(acc, P * P, rs->tl.P[wave]);
                    if (lane == 0) st_sc1(qv.vals + e, granule(tag, ent));
                }
                mine += got;
                if (tid == 0) post(nextc);
                __syncthreads();
            }
            // the rest is with other waves of the launch (or done): poll the result slots
            CGIC_RQ_STAMP(qid, 1, wall_clock64());
            CGIC_RQ_STAMP(qid, 4, mine); (void)mine;
            drain_stores();               // (this workgroup's own results)
#pragma unroll 1
            for (;;) {
                if (tid == 0) rs->flag = 0;
                __syncthreads();
                unsigned int missing = 0;
                for (unsigned int k = tid; k < Lq; k += NT)
                    missing += (unsigned int)(ld_sc1(qv.vals + qitems[k]) >> 32) != tag ? 1u : 0u;
                if (missing) rs->flag = 1;
                __syncthreads();
                const unsigned int again = rs->flag;
                __syncthreads();
                if (!again) break;
                __builtin_amdgcn_s_sleep(4);
            }
            CGIC_RQ_STAMP(qid, 2, wall_clock64());
            if (tid == 0) cnt[4] = 0;
            for (int k = tid; k < L; k += NT) {
                const int e = (int)lst[k];
                if (todo(k, e)) out(k, e, __uint_as_float((unsigned int)ld_sc1(qv.vals + e)));
            }
        } else if (P == 16 && L * UPP <= NWR) {
            // a couple of 16x16 patches: four waves each, side by side
            const int e = wave < L * UPP ? (int)lst[wave / UPP] : -1;
            const bool go = e >= 0 && todo(wave / UPP, e);
            if (go) unit(e, wave % UPP, rs->tl.T[wave]);
            __syncthreads();
            if (go && wave % UPP == 0) {
                float acc = 0.f;
                for (int q = 0; q < UPP; ++q) acc = ref_add_rows(acc, rs->tl.T[wave + q]);
                const float ent = ref_finalize(acc, P * P, rs->tl.P[wave]);
                if (lane == 0) out(wave / UPP, e, ent);
            }
        } else if (wave < NWR) {
            for (int k = wave; k < L; k += NWR) {
                const int e = (int)lst[k];
                if (!todo(k, e)) continue;                            // (wave-uniform)
                float acc = 0.f;
                for (int q = 0; q < UPP; ++q) {
                    unit(e, q, rs->tl.T[wave]);
                    acc = ref_add_rows(acc, rs->tl.T[wave]);
                }
                const float ent = ref_finalize(acc, P * P, rs->tl.P[wave]);
                if (lane == 0) out(k, e, ent);
            }
        }
        __syncthreads();
    };

    const float *flat = rf.flat8 ? rf.flat8 + img0 * (P == 16 ? 4 * (int64_t)nP : (int64_t)nP) : nullptr;     // this segment's 8x8 flags
    auto flat_key_of = [&](int i) -> unsigned int {        // gray bits of constant patch i, or kRefEmpty
        if (P == 8) { const float f = flat[i]; return f == f ? __float_as_uint(f) : kRefEmpty; }
        const int bi = i / nP, r = i - bi * nP, py = r / wP, px = r - py * wP;
        const float *q = flat + (int64_t)bi * 4 * nP + (int64_t)(2 * py) * (2 * wP) + 2 * px;
        const float f = q[0];
        return (f == q[1] && f == q[2 * wP] && f == q[2 * wP + 1]) ? __float_as_uint(f) : kRefEmpty;           // (NaN: never equal)
    };

    // the same in two steps, for passes that load a batch of flags before they look at any: what to load (four flags of a 16x16
    // patch; i < n), then the key
    auto flat_raw = [&](int i, float (&r)[P == 16 ? 4 : 1]) {
        if (P == 8) { r[0] = flat[i]; return; }
        const int bi = i / nP, q = i - bi * nP, py = q / wP, px = q - py * wP;
        const float *f = flat + (int64_t)bi * 4 * nP + (int64_t)(2 * py) * (2 * wP) + 2 * px;
        r[0] = f[0]; r[1] = f[1]; r[2] = f[2 * wP]; r[3] = f[2 * wP + 1];
    };
    auto flat_key_raw = [&](const float (&r)[P == 16 ? 4 : 1]) -> unsigned int {
        if (P == 8) return r[0] == r[0] ? __float_as_uint(r[0]) : kRefEmpty;
        return (r[0] == r[1] && r[0] == r[2] && r[0] == r[3]) ? __float_as_uint(r[0]) : kRefEmpty;
    };

    const bool few = m_all <= 64 && rank >= c_below && rank - c_below < m_all;
    // FEW: constant patches of one gray are one evaluation: member k is evaluated only if it is the first of its gray.  Every wave
    // works that out for itself (lane = member): no LDS, no barrier
    unsigned int key = kRefEmpty;
    int my = 0, lead = lane;
    bool exact_l = true;
    unsigned long long work = 0;
    if (__builtin_expect(few, 1)) {
        my = lane < (int)m_all ? (int)rs->small[lane] : 0;
        exact_l = lane < (int)m_all ? is_exact(my) : true;
        if (flat && lane < (int)m_all && !exact_l) key = flat_key_of(my);
        // (the member with the smallest PATCH INDEX of its gray: the same one in every row band, whatever order they listed them in)
        int lead_e = my;
        for (int j = 0; j < (int)m_all; ++j) {
            const unsigned int kj = (unsigned int)__builtin_amdgcn_readlane((int)key, j);
            const int ej = __builtin_amdgcn_readlane(my, j);
            if (kj == key && key != kRefEmpty && ej < lead_e) { lead = j; lead_e = ej; }
        }
        work = __ballot(lane < (int)m_all && !exact_l && lead == lane);      // bit k: member k is evaluated
    }
    if constexpr (BAIL) {
        // what there is to evaluate from pixels: FEW -- the members that are not a follower of their gray; MANY -- counted above.
        // Up to four rounds of this workgroup's waves stay here (the restart costs as much); constant patches do not count: their
        // passes cost the same in both instantiations.
        const unsigned int heavy = few ? (unsigned int)__builtin_popcountll(work) : cnt[7];
        // (row bands: more than four rounds, as before; an image with a workgroup of its own (round 6: the launch's queues): more than
        // eight -- the restart and the queue's hand-offs cost ~15 us, and a band of a few rounds is done sooner where it was found)
        // (... and only a band of at most 64 members, the FEW form: a band of hundreds of mostly constant patches -- flat regions with
        // edges -- spends its time in passes over the map that the restart would repeat: 57 us where it was found, 62 with the queues)
#ifdef CGIC_FUSED_Q_MANY        // dev A/B: long bands of more than 64 members of an image with a workgroup of its own go to the queues too
        const bool many_ok = true;
#else
        const bool many_ok = false;
#endif
#ifndef CGIC_FUSED_Q_ROUNDS
#define CGIC_FUSED_Q_ROUNDS 8
#endif
        if (a.rq.nq != 0 && heavy * UPP > (nb > 1 ? 4 : CGIC_FUSED_Q_ROUNDS) * NWR && (nb > 1 || few || many_ok)) {      // (workgroup-uniform, and the same in every row band)
            *bail_out = true;
            __builtin_amdgcn_s_setprio(0);
            return t_a;
        }
    }
    // Row bands: does this refinement go to the queue?  (The same answer in every band: the counts do not depend on the order in
    // which a band's threads listed the members.)  Then the first band to arrive does it for all.
    if (RQ && nb > 1 && have_q && (!few || __builtin_popcountll(work) * UPP > 4 * NWR)) {
        if (tid == 0) rs->flag = add_sc1(qv.hdr + QH_ARRIVE, 1u);
        __syncthreads();
        share = rs->flag == 0;
        if (!share) {
            // not the owner: its waves evaluate what the owner publishes (or anything else) until the outcome is there, then take it over
            if (wave < NWR) {
#pragma unroll 1
                for (;;) {
                    if (refine_help_wave(rc, rs->tl.T[wave], rs->tl.rec[wave], rs->tl.P[wave], rs->tl.bins, qid, blockIdx.x * 16u + (unsigned int)wave)) continue;
                    if (ld_sc1(qv.hdr + QH_FIN)) break;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            __syncthreads();
            const unsigned int nf = ld_sc1(qv.hdr + QH_NFINAL);
            const float thr = __uint_as_float(ld_sc1(qv.hdr + QH_THR));
            for (unsigned int k = tid; k < nf; k += NT) arr[ld_sc1(qv.fidx + k)] = ld_sc1(qv.fval + k);
            __syncthreads();
            if (tid == 0 && add_sc1(qv.hdr + QH_LEFT, 1u) == (unsigned int)nb - 1u) {       // the last band out restores the header
                st_sc1(qv.hdr + QH_ARRIVE, 0u); st_sc1(qv.hdr + QH_FIN, 0u); st_sc1(qv.hdr + QH_LEFT, 0u);
                st_sc1(qv.hdr + QH_THR, 0u); st_sc1(qv.hdr + QH_NFINAL, 0u);
            }
            __builtin_amdgcn_s_setprio(0);
            return thr;
        }
    }
    // the owner's hand-over to the other row bands: the final list is complete in qv.fidx / qv.fval (nf entries)
    auto hand_over = [&](unsigned int nf, float thr) {
        drain_stores();
        __syncthreads();
        if (tid == 0) {
            st_sc1(qv.hdr + QH_NFINAL, nf);
            st_sc1(qv.hdr + QH_THR, __float_as_uint(thr));
            drain_stores();
            st_sc1(qv.hdr + QH_FIN, 1u);
            if (add_sc1(qv.hdr + QH_LEFT, 1u) == (unsigned int)nb - 1u) {
                st_sc1(qv.hdr + QH_ARRIVE, 0u); st_sc1(qv.hdr + QH_FIN, 0u); st_sc1(qv.hdr + QH_LEFT, 0u);
                st_sc1(qv.hdr + QH_THR, 0u); st_sc1(qv.hdr + QH_NFINAL, 0u);
            }
        }
    };

    if (__builtin_expect(few, 1)) {
        // (a band that one round of this workgroup's waves evaluates is not worth the exchange -- two global round trips: every row band does it all)
        const bool sf = split && __builtin_popcountll(work) * UPP > NWR;
        sweep(rs->small, (int)m_all, __builtin_popcountll(work), [&](int k, int e) { return ((work >> k) & 1ull) && (!sf || mine_of(e)); },
              [&](int, int e, float ent) { arr[e] = ent; if (sf) st_sc1(xvals + e, ent); });
        if (__builtin_expect(sf, 0)) {
            const unsigned long long own = __ballot(lane < (int)m_all && ((work >> lane) & 1ull) && mine_of(my));
            exchange((unsigned int)__builtin_popcountll(work), (unsigned int)__builtin_popcountll(own));
            if (wave == 0 && lane < (int)m_all && ((work >> lane) & 1ull) && !mine_of(my)) arr[my] = ld_sc1(xvals + my);
            exchange_leave();
        }
        // everything below the band is below the true threshold, everything above it above: it is the band's (rank - below)-th
        if (wave == 0) {
            float v = lane < (int)m_all ? arr[rs->small[lead]] : __builtin_inff();        // (a follower takes its leader's value)
            if (lane < (int)m_all && lead != lane) arr[my] = v;
            if (share && lane < (int)m_all) { st_sc1(qv.fidx + lane, (unsigned int)my); st_sc1(qv.fval + lane, v); }
            unsigned int less = 0, leq = 0;
            for (unsigned int j = 0; j < m_all; ++j) {
                const float vj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)j));
                less += vj < v ? 1u : 0u;
                leq += vj <= v ? 1u : 0u;
            }
            const unsigned int r = rank - c_below;
            if (lane < (int)m_all && less <= r && r < leq) rs->thr = v;       // (lanes holding equal values write the same bits)
        }
        __syncthreads();
        const float thr = rs->thr;
        if (share) hand_over(m_all, thr);
        __builtin_amdgcn_s_setprio(0);
        return thr;
    }

    // ---- MANY.  Membership of the band is decided on the values as they were when the select ran: every element is tested
    // before it is written, and written once.
    // Three things made this path slow on flat content (a 768x768 tile with 6408 constant patches in the medium band spent 54 us
    // here, none of it evaluating): every pass over the map read the constant-patch flags one dependent global load per element
    // and thread; whole waves inserted the same gray into the table (64 lanes on one LDS word); and the members were listed
    // 1024 indices at a time, three barriers per range.  Now: a pass loads a batch of values AND flags before it looks at any;
    // a wave whose members are all one gray inserts it once; the members that need their pixels go into a bitmap (a wave's 64
    // consecutive elements are one word: a ballot, no atomics), which also puts them in index order -- the same order in every
    // row band of a tile, so that the bands can take every nb-th one (SPLIT) -- and is listed 1024 MEMBERS at a time.
    auto put = [&](int i, float v) {
        arr[i] = v;
        if (share) { const unsigned int pos = atomicAdd(&cnt[5], 1u); st_sc1(qv.fidx + pos, (unsigned int)i); st_sc1(qv.fval + pos, v); }
    };
    if constexpr (!(SPLIT || RQ)) {
        // (the fused launch's plain instantiation keeps the round-4 form of this path: the one below cost its ORDINARY path 2 us
        // through register allocation, B = 64 x 256x256 24.0 -> 26.4 us, like every other addition to that kernel)
        auto in_band = [&](int i) { return fabsf(arr[i] - t_a) <= w && !is_exact(i); };
        if (flat) {
            // the distinct grays of the band's constant patches; the member with the smallest index stands for its gray
            unsigned int *flat_rep = rs->list;               // [kRefFlatSlots] (the list is free until the ranges below)
            for (int i = tid; i < kRefFlatSlots; i += NT) { rs->flat_key[i] = kRefEmpty; flat_rep[i] = kRefEmpty; }
            __syncthreads();
            for (int i = tid; i < n; i += NT)
                if (in_band(i)) {
                    const unsigned int k = flat_key_of(i);
                    if (k != kRefEmpty) { const int slot = flat_insert(rs->flat_key, k); if (slot >= 0) atomicMin(&flat_rep[slot], (unsigned int)i); }
                }
            __syncthreads();
            unsigned int used = 0;
            for (int i = lane; i < kRefFlatSlots; i += 64) used += rs->flat_key[i] != kRefEmpty ? 1u : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) used += __shfl_xor(used, d, kWave);
            if (RQ && have_q && (int)used * UPP > 4 * NWR) {
                // (evaluated from the pixels like any other patch -- 64 or 256 equal ones: the same bits as one evaluation of the gray)
                sweep(flat_rep, kRefFlatSlots, (int)used, [&](int, int e) { return (unsigned int)e != kRefEmpty; },
                      [&](int k, int, float ent) { rs->flat_val[k] = ent; });
            } else {
                if (wave < NWR) {
                    for (int s = wave; s < kRefFlatSlots; s += NWR) {
                        const unsigned int k = rs->flat_key[s];              // (wave-uniform)
                        if (k == kRefEmpty) continue;
                        int j0;
                        float v[kRefWin];
                        ref_pixel(rs->tl.bins, rf.sigma, __uint_as_float(k), j0, v);        // every pixel of the patch is this one
                        ref_unit_chunks(rs->tl.rec[wave], j0, v, rs->tl.T[wave]);
                        float acc = 0.f;
                        for (int q = 0; q < UPP; ++q) acc = ref_add_rows(acc, rs->tl.T[wave]);
                        const float ent = ref_finalize(acc, P * P, rs->tl.P[wave]);
                        if (lane == 0) rs->flat_val[s] = ent;
                    }
                }
                __syncthreads();
            }
        }
        for (int base = 0; base < n; base += kRefListCap) {
            if (tid == 0) cnt[2] = 0;
            __syncthreads();
            const int hi = base + kRefListCap < n ? base + kRefListCap : n;
            for (int i = base + tid; i < hi; i += NT) {
                if (!in_band(i)) continue;
                int slot = -1;
                if (flat) { const unsigned int k = flat_key_of(i); if (k != kRefEmpty) slot = flat_find(rs->flat_key, k); }
                if (slot >= 0) put(i, rs->flat_val[slot]);
                else rs->list[atomicAdd(&cnt[2], 1u)] = (unsigned int)i;
            }
            __syncthreads();
            const int L = (int)cnt[2];
            sweep(rs->list, L, L, [](int, int) { return true; }, [&](int, int e, float ent) { put(e, ent); });
        }
    } else {
        CGIC_RS_STAMP(0);
        const int W64 = (n + 63) >> 6;
        unsigned int *flat_rep = rs->list;               // [kRefFlatSlots] (the list is free until the members are listed)
        if (flat) {
            for (int i = tid; i < kRefFlatSlots; i += NT) { rs->flat_key[i] = kRefEmpty; flat_rep[i] = kRefEmpty; }
            __syncthreads();
        }
        constexpr int U = 4;
        // pass A: constant members -> the table of distinct grays (the member with the smallest index stands for its gray), the others -> the bitmap
        for (int i0 = tid; i0 < 64 * W64; i0 += U * NT) {
            float v[U];
            float r[U][P == 16 ? 4 : 1];
            unsigned int k[U];
            // (unconditional loads from clamped indices, nothing looked at: a load inside a branch waits inside the branch, one
            // global round trip per element -- what made these passes 10-14 us each)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * NT, ic = i < n ? i : n - 1;
                v[u] = arr[ic];
                if (flat) flat_raw(ic, r[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * NT;                 // (a wave's lanes: 64 consecutive elements, one word of the bitmap; uniform trip count)
                if (i - lane >= 64 * W64) continue;        // (no break: the loop stays unrolled, the arrays stay registers)
                k[u] = (flat && i < n) ? flat_key_raw(r[u]) : kRefEmpty;
                const bool member = i < n && fabsf(v[u] - t_a) <= w && !is_exact(i < n ? i : 0);
                bool need = member && k[u] == kRefEmpty;
                if (member && k[u] != kRefEmpty) {
                    // (regions of one gray: the wave inserts it once)
                    const unsigned int k0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)k[u]);
                    const unsigned long long act = __ballot(true);
                    int slot;
                    if (__ballot(k[u] == k0) == act) {
                        slot = 0;
                        if (lane == __builtin_ctzll(act)) { slot = flat_insert(rs->flat_key, k0); if (slot >= 0) atomicMin(&flat_rep[slot], (unsigned int)i); }
                        slot = __builtin_amdgcn_readfirstlane(slot);
                    } else {
                        slot = flat_insert(rs->flat_key, k[u]);
                        if (slot >= 0) atomicMin(&flat_rep[slot], (unsigned int)i);
                    }
                    need = slot < 0;                       // (table full: see below)
                    if (slot < 0) cnt[6] = 1u;
                }
                const unsigned long long word = __ballot(need);
                if (lane == 0) rs->bits[i >> 6] = word;
            }
        }
        __syncthreads();
        // More distinct grays than the table holds (a blocky image whose quantile falls among its constant blocks): WHICH grays made
        // it into the table depends on the order the waves came by -- not the same in every row band, and the bands must agree on
        // the member list.  Then no gray is shared: every member is evaluated from its pixels.
        const bool overflow = flat && cnt[6] != 0u;
        if (overflow) {
            for (int i0 = tid; i0 < 64 * W64; i0 += NT) {
                const bool member = i0 < n && fabsf(arr[i0 < n ? i0 : 0] - t_a) <= w && !is_exact(i0 < n ? i0 : 0);
                const unsigned long long word = __ballot(member);
                if (lane == 0) rs->bits[i0 >> 6] = word;
            }
            __syncthreads();
        }
        CGIC_RS_STAMP(1);
        if (flat && !overflow) {
            unsigned int used = 0;
            for (int i = lane; i < kRefFlatSlots; i += 64) used += rs->flat_key[i] != kRefEmpty ? 1u : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) used += __shfl_xor(used, d, kWave);
            if (RQ && have_q && (int)used * UPP > 4 * NWR) {
                // (evaluated from the pixels like any other patch -- 64 or 256 equal ones: the same bits as one evaluation of the gray)
                sweep(flat_rep, kRefFlatSlots, (int)used, [&](int, int e) { return (unsigned int)e != kRefEmpty; },
                      [&](int k, int, float ent) { rs->flat_val[k] = ent; });
            } else if (used) {
                if (wave < NWR) {
                    for (int s = wave; s < kRefFlatSlots; s += NWR) {
                        const unsigned int k = rs->flat_key[s];              // (wave-uniform)
                        if (k == kRefEmpty) continue;
                        int j0;
                        float v[kRefWin];
                        ref_pixel(rs->tl.bins, rf.sigma, __uint_as_float(k), j0, v);        // every pixel of the patch is this one
                        ref_unit_chunks(rs->tl.rec[wave], j0, v, rs->tl.T[wave]);
                        float acc = 0.f;
                        for (int q = 0; q < UPP; ++q) acc = ref_add_rows(acc, rs->tl.T[wave]);
                        const float ent = ref_finalize(acc, P * P, rs->tl.P[wave]);
                        if (lane == 0) rs->flat_val[s] = ent;
                    }
                }
                __syncthreads();
            }
            CGIC_RS_STAMP(2);
            // pass B: the constant members take their gray's value (the others are still untouched: the band test sees what the select saw)
            if (used) {
                for (int i0 = tid; i0 < n; i0 += U * NT) {
                    float v[U];
                    float r[U][P == 16 ? 4 : 1];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = i0 + u * NT, ic = i < n ? i : n - 1;
                        v[u] = arr[ic];
                        flat_raw(ic, r[u]);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = i0 + u * NT;
                        const unsigned int ku = i < n ? flat_key_raw(r[u]) : kRefEmpty;
                        if (ku == kRefEmpty || !(fabsf(v[u] - t_a) <= w) || is_exact(i)) continue;
                        const int slot = flat_find(rs->flat_key, ku);
                        if (slot >= 0) put(i, rs->flat_val[slot]);
                    }
                }
                __syncthreads();
            }
        }
        // the members in index order: how many in front of each word
        if (wave == 0) {
            unsigned int carry = 0;
            for (int c = 0; c < W64; c += 64) {
                const unsigned int pc = c + lane < W64 ? (unsigned int)__builtin_popcountll(rs->bits[c + lane]) : 0u;
                const unsigned int incl = wave_inclusive_scan_u32(pc);
                if (c + lane < W64) rs->pre[c + lane] = carry + incl - pc;
                carry += (unsigned int)__builtin_amdgcn_readlane((int)incl, 63);
            }
            if (lane == 0) cnt[2] = carry;
        }
        __syncthreads();
        const unsigned int M = cnt[2];                               // members evaluated from their pixels, all row bands together
        // SPLIT: member number o (in index order) is band (o mod nb)'s, its (o / nb)-th.  (o < 2^14, nb <= 8: the multiply is exact)
        const unsigned int nbu = split ? (unsigned int)nb : 1u, inv = ((1u << 17) + nbu - 1u) / nbu;
        const unsigned int own = M > (unsigned int)band * (split ? 1u : 0u) ? (split ? ((M - (unsigned int)band - 1u) * inv >> 17) + 1u : M) : 0u;
        CGIC_RS_STAMP(3);
        for (unsigned int c0 = 0; c0 < own; c0 += kRefListCap) {
            for (int wd = tid; wd < W64; wd += NT) {
                unsigned long long b = rs->bits[wd];
                unsigned int o = rs->pre[wd];
                while (b) {
                    const int bit = __builtin_ctzll(b);
                    b &= b - 1;
                    const unsigned int q = o * inv >> 17;
                    if (o - q * nbu == (split ? (unsigned int)band : 0u) && q - c0 < (unsigned int)kRefListCap) rs->list[q - c0] = (unsigned int)(64 * wd + bit);
                    ++o;
                }
            }
            __syncthreads();
            const int L = (int)(own - c0 < (unsigned int)kRefListCap ? own - c0 : (unsigned int)kRefListCap);
            sweep(rs->list, L, L, [](int, int) { return true; }, [&](int, int e, float ent) { put(e, ent); if (split) st_sc1(xvals + e, ent); });
        }
        CGIC_RS_STAMP(4);
        if (split && M) {
            exchange(M, own);
            CGIC_RS_STAMP(5);
            for (int wd = tid; wd < W64; wd += NT) {                 // the other bands' members
                unsigned long long b = rs->bits[wd];
                unsigned int o = rs->pre[wd];
                while (b) {
                    const int bit = __builtin_ctzll(b);
                    b &= b - 1;
                    const unsigned int q = o * inv >> 17;
                    if (o - q * nbu != (unsigned int)band) arr[64 * wd + bit] = ld_sc1(xvals + 64 * wd + bit);
                    ++o;
                }
            }
            exchange_leave();
        }
        CGIC_RS_STAMP(6);
    }
    const lds_cf32 arr_l = as_lds(arr);                  // (refinement only runs on staged maps)
    const float thr = radix_select<NT>([&](int64_t i) { return arr_l[i]; }, n, rank, sh);
    CGIC_RS_STAMP(7);
    if (share) { __syncthreads(); hand_over(cnt[5], thr); }
    __builtin_amdgcn_s_setprio(0);
    return thr;
}

// The whole router for segment `seg`, executed by a block of NT threads; `dyn` = dynamic LDS of at least router_lds_bytes() bytes.
// ST: both maps are staged in LDS (a.stage == 1: every segment that fits -- per-image routing up to 768x768 tiles): the copy the
// launches of the timed paths run, all element reads ds_read; the other copy serves the unstaged / half-staged segments.
// HELP: the stand-alone launch -- with the refinement queues (refine_select's RQ), and a router workgroup that is done evaluates
// other images' band patches while any is open.  The fused launch's routers refine inside their own workgroup only.
// SPLIT: the row bands of a tile split a threshold band between them (refine_select) -- its own instantiation, entered only as a
// launch's second attempt (router_body): even this much more code in the router's one path costs its ordinary path 2-3 us
// (B = 64 x 256x256: 23.4 -> 26.5 us).  BAIL: the first attempt of such a launch (returns true when it left for the second).
template <int NT, bool ST, bool HELP, bool SPLIT, bool BAIL>
__device__ __forceinline__ int router_team(const RouterArgs &a, int64_t blk, unsigned char *dyn, unsigned int pre_held = 0, RouterResume *res = nullptr);

// A SPLIT launch runs the router in two attempts: first the plain instantiation, which leaves (BAIL) as soon as a select's band is
// more than four rounds of work -- before anything but the coarse mask is written, and that is written again with the same values --
// then, from the top, the instantiation in which the row bands split the band.  The ordinary tile never enters the second copy, and
// nothing of the first is live across it: its path keeps the plain kernel's register allocation (with the split code inlined
// into the one path an ordinary 768x768 tile paid 3.5 us, the 2040x1356 chain 8 us; NOTES 11.8).  A tile that does refine at
// length pays the first attempt on top (7-20 us of 70-90).
// where router_team keeps its RefineShared (stage 1, refinement on): behind the gate bits and the two staged maps
__device__ __forceinline__ RefineShared *router_refine_shared(const RouterArgs &a, unsigned char *dyn)
{
    const int64_t N16 = a.per * a.h16 * a.w16, N8 = 4 * N16;
    unsigned long long *gc_bits = reinterpret_cast<unsigned long long *>(dyn + kRouterSharedBytes);
    float *l16 = reinterpret_cast<float *>(gc_bits + ((N16 + 63) >> 6));
    return reinterpret_cast<RefineShared *>((reinterpret_cast<uintptr_t>(l16 + N16 + N8) + 15) & ~(uintptr_t)15);
}

template <int NT, bool HELP = false, bool SPLIT = false>
__device__ __forceinline__ void router_body(const RouterArgs &a, int64_t blk, unsigned char *dyn)
{
    if (a.stage == 1) {
        if constexpr (SPLIT) {
            RouterResume rr;
            rr.at = 0; rr.thr = 0.f; rr.si.h = nullptr; rr.si.digit = 0; rr.si.before = 0;
#ifdef CGIC_FUSED_Q_NORESUME      // dev A/B: the second attempt starts from the top
            const int how = router_team<NT, true, false, false, true>(a, blk, dyn);
#else
            const int how = router_team<NT, true, false, false, true>(a, blk, dyn, 0u, &rr);
#endif
            if (how != 1) {
                // Done in the plain code.  One image per workgroup (round 6): the images of the launch whose bands ARE long have
                // started over with the refinement queues -- while any of them holds a band open, the waves of a workgroup that had
                // a (short) band of its own evaluate patches for them instead of leaving: tie-heavy content comes by the batch.  The
                // ordinary image (how == 0: no band) leaves at once -- the look at the board is a memory round trip at the very end
                // of the launch's critical path (measured: +1.9 us on every launch when every router took it).
#ifndef CGIC_FUSED_Q_NOAFTER
#ifdef CGIC_FUSED_Q_REFINERS_ONLY      // dev A/B: only the routers that refined look at the board (the form of the first half of round 6)
                if (how == 2 && a.bands <= 1 && a.rq.nq != 0 && a.rf.x != nullptr) {
#else
                // (a launch WITH queues is one the caller chose for tie-heavy batches -- pipeline.HotPathPipeline.decide: there every
                // router looks, and an image that starts over says so on the board BEFORE it does (below): the ordinary images'
                // routers are done 2-3 us before its list is out and used to be gone by then -- smooth 8-bit batches: two images of
                // 64 with 30-40 patches each and ~12 helpers that arrived 5-15 us late)
                if (a.bands <= 1 && a.rq.nq != 0 && a.rf.x != nullptr) {
#endif
                    RefineShared *rs = router_refine_shared(a, dyn);
                    __syncthreads();
                    if (threadIdx.x == 0) rs->flag = ld_sc1(a.rq.board + QB_BUSY);
                    __syncthreads();
#ifndef CGIC_FUSED_HELP_WAVES
#define CGIC_FUSED_HELP_WAVES 2
#endif
                    if (rs->flag) refine_help_while_busy(a, &rs->tl, how == 2 ? kRefWavesMax : CGIC_FUSED_HELP_WAVES);
                }
#endif
                return;
            }
            __syncthreads();
            // second attempt, from the top: the row bands of a large tile split the band between them; an image with a workgroup of
            // its own publishes it to the launch's queues (the stand-alone launch's instantiation: owner + helpers)
#ifdef CGIC_FUSED_Q_NOHELP      // dev A/B: without the queue instantiation in the kernel
            router_team<NT, true, false, true, false>(a, blk, dyn);
#else
            if (a.bands > 1) router_team<NT, true, false, true, false>(a, blk, dyn);
            else {
#ifndef CGIC_FUSED_Q_REFINERS_ONLY
                if (threadIdx.x == 0) add_sc1(a.rq.board + QB_BUSY, 1u);        // held until this image's last select (router_team: busy_held)
                router_team<NT, true, true, false, false>(a, blk, dyn, 1u, &rr);
#else
                router_team<NT, true, true, false, false>(a, blk, dyn);
#endif
            }
#endif
        } else {
            router_team<NT, true, HELP, false, false>(a, blk, dyn);
        }
    } else {
        router_team<NT, false, false, false, false>(a, blk, dyn);
    }
}

// (returns 1 when it left early -- BAIL only --, 2 when it finished and one of its selects evaluated a band from the pixels, else 0)
template <int NT, bool ST, bool HELP, bool SPLIT, bool BAIL>
__device__ __forceinline__ int router_team(const RouterArgs &a, int64_t blk, unsigned char *dyn, unsigned int pre_held, RouterResume *res)
{
    const int nb = a.bands > 1 ? a.bands : 1;
    const int64_t seg = blk / nb;
    const int band = (int)(blk - seg * nb);
    RouterShared *sh = reinterpret_cast<RouterShared *>(dyn);
    unsigned long long *gc_bits = reinterpret_cast<unsigned long long *>(dyn + kRouterSharedBytes);  // [ceil(N16/64)]

    const int tid = threadIdx.x;
    const int lane = lane_id();
    CGIC_STAMP(0);
#ifdef CGIC_PHASE_CLOCKS      // dev: (start, after coarse select, after medium select, end) of the first 64 router workgroups: g_blk_t slots 1024 + 2 blk ..
#define CGIC_RT_STAMP(k) do { if (threadIdx.x == 0 && blk < 64) g_blk_t[2 * (1024 + 2 * blk) + (k)] = wall_clock64(); } while (0)
#else
#define CGIC_RT_STAMP(k) do {} while (0)
#endif
    CGIC_RT_STAMP(0);
    const int64_t h16 = a.h16, w16 = a.w16, h8 = 2 * h16, w8 = 2 * w16, h4 = 4 * h16, w4 = 4 * w16;
    const int64_t n16 = h16 * w16, n8 = h8 * w8, n4 = h4 * w4;
    const int64_t N16 = a.per * n16, N8 = a.per * n8, N4 = a.per * n4;
    const float *e16 = a.e16 + seg * N16;
    const float *e8 = a.e8 + seg * N8;
    RefineShared *rs = nullptr;
    // (second attempt after a first one that left at its coarse select: the maps' copies, that select's last histogram and hist[0],
    // cleared by its last pass, are as it left them)
    const bool resume = !BAIL && ST && res != nullptr && res->at == 1;
    if (!resume) for (int i = tid; i < 256; i += NT) sh->hist[0][i] = 0;        // (radix_select: the first select's first buffer)
    if (!a.stage) __syncthreads();
    if (a.stage) {
        // one round trip to HBM/L2 instead of one per radix pass (8 passes + 3 elementwise sweeps)
        float *l16 = reinterpret_cast<float *>(gc_bits + ((N16 + 63) >> 6));
        float *l8 = a.stage == 1 ? l16 + N16 : l16;               // stage 2: e16 stays in global memory
        if (a.stage == 1) {
            if (!resume) for (int64_t i = tid; i < N16; i += NT) l16[i] = e16[i];
            e16 = l16;
        }
        if (!resume) for (int64_t i = tid; i < N8; i += NT) l8[i] = e8[i];
        e8 = l8;
        if (a.rf.x) {
            rs = reinterpret_cast<RefineShared *>((reinterpret_cast<uintptr_t>(l8 + N8) + 15) & ~(uintptr_t)15);
            if (tid < 16 && !(resume && tid < 8)) rs->cnt[tid] = 0;       // (resume: the coarse select's counts stay, see refine_select's `precounted`)
            if (tid == 0) rs->busy_held = pre_held;          // (router_body: a second attempt announced itself on the board before it started over)
            if (tid < kBins) rs->tl.bins[tid] = linspace_bin(tid);
        }
        __syncthreads();
    }
    const bool refine = rs != nullptr;          // (the host only asks for it at stage 1: both maps in LDS)
    // element reads: ds_read in the staged copy (see lds_cf32)
    auto rd16 = [&](int64_t i) -> float { if constexpr (ST) return as_lds(e16)[i]; else return e16[i]; };
    auto rd8 = [&](int64_t i) -> float { if constexpr (ST) return as_lds(e8)[i]; else return e8[i]; };
    CGIC_STAMP(1);
    int32_t *mc = a.mask_c + seg * N16;
    int32_t *mm = a.mask_m + seg * N8;
    int32_t *mf = a.mask_f + seg * N4;
    const int mode = a.mode;
    const bool has_thr_c = mode == 0 || mode == 2 || mode == 3;
    bool bail = false, refined = false;

    // ---- coarse gate (RouterTriple.py:21-25 / 52-56 / 63-66)
    float thr_c = 0.f;
    if (has_thr_c) {
        SelInfo si;
        if (resume) { thr_c = res->thr; si = res->si; }
        else thr_c = radix_select<NT>(rd16, N16, a.rank_c, sh, &si);
        if constexpr (ST) if (refine)
            thr_c = refine_select<NT, 16, HELP, SPLIT, BAIL>(a, (int)(2 * seg), nb, band, const_cast<float *>(e16), (int)N16, a.rank_c, thr_c, ExactGate{nullptr, 0, 0, 0, 0, 0, 0},
                                          seg * a.per, (int)w16, (int)n16, rs, sh, si, &bail, &refined, resume);
        if (BAIL && bail) {
            if (res) { res->at = 1; res->thr = thr_c; res->si = si; }       // (refine_select returned the approximate threshold as it got it)
            return 1;
        }
    }
    CGIC_STAMP(2);
    CGIC_RT_STAMP(1);
    const int64_t N16r = (N16 + 63) & ~(int64_t)63;
    for (int64_t i = tid; i < N16r; i += NT) {
        bool g = false;
        if (i < N16) g = has_thr_c ? (rd16(i) < thr_c) : (mode == 4);
        unsigned long long bal = __ballot(g);
        if (lane == 0) gc_bits[i >> 6] = bal;
        if (i < N16 && band == 0) mc[i] = g ? 1 : 0;
    }
    // (both maps staged, mode 0: the sweep that masks the medium map takes the coarse gate from the coarse map itself, not from
    // these bits -- nothing reads them before that sweep's barrier)
    if (!(ST && mode == 0)) __syncthreads();
    // 32-bit index math throughout (N8 < 2^31 is checked on the host): a 64-bit divide is ~100 instructions
    const int n8i = (int)n8, w8i = (int)w8, n16i = (int)n16, w16i = (int)w16;
    // n / d as one v_mul_hi_u32 with m = ceil(2^32 / d): exact while n * d < 2^32 (n < 2^31 / 4 here, d <= 2^13: the host checks);
    // a 32-bit divide is ~25 instructions, and the sweeps below do two per element
    const unsigned int mg_n8 = a.mg_n8, mg_w8 = a.mg_w8, mg_n4 = a.mg_n4, mg_w4 = a.mg_w4;
    auto fdiv = [](int n, unsigned int m, int d) -> int { return m ? (int)__umulhi((unsigned int)n, m) : n / d; };
    auto gc_at = [&](int b, int y8, int x8) -> bool {   // coarse gate of the parent of medium element (y8, x8) of image b
        const int c = b * n16i + (y8 >> 1) * w16i + (x8 >> 1);
        return (gc_bits[c >> 6] >> (c & 63)) & 1ull;
    };
    auto gc_of8 = [&](int64_t i) -> bool {
        const int ii = (int)i;
        const int b = fdiv(ii, mg_n8, n8i), r = ii - b * n8i;
        const int y = fdiv(r, mg_w8, w8i), x = r - y * w8i;
        return gc_at(b, y, x);
    };
    // One image per segment (per-image routing: every batched compress): walk rows by wave and columns by lane -- the
    // flat loops below spend ~4 integer divisions (~25 VALU instructions each) per element, which at 9216 + 36864
    // elements per 768x768 tile on ONE CU was 11 of the router's 39 us.
    const bool rows2d = a.per == 1 && (w8 >= 64 || nb > 1);      // (narrow rows leave most lanes of a wave idle: flat loops there)
    constexpr int NWV = NT / 64;
    const int wv = tid >> 6;
    // this workgroup's band of coarse rows [cy0, cy1) (medium rows x2, fine rows x4); bands > 1 only with rows2d
    const int cper = ((int)h16 + nb - 1) / nb;
    const int cy0 = band * cper < (int)h16 ? band * cper : (int)h16;
    const int cy1 = cy0 + cper < (int)h16 ? cy0 + cper : (int)h16;

    CGIC_STAMP(3);
    // ---- medium gate
    float thr_m = 0.f;
    if (mode == 0) {      // :27-31: sort e8 * (1 - up2(gate_coarse))
        if (a.stage) {
            // mask the LDS copy IN PLACE, once, so the four radix passes are plain LDS sweeps.  (A second, masked copy used to
            // sit beside the plain one; the plain value of a gated element is never needed again: its medium gate is
            // (v < thr) && !gate_coarse whatever v is -- 4 N8 bytes of LDS less.)
            float *l8m = const_cast<float *>(e8);
            __attribute__((address_space(3))) float *l8w = (__attribute__((address_space(3))) float *)l8m;
            SelInfo si;
            if constexpr (ST) {
                // One sweep: the parent's gate straight from the coarse map (the same comparison the gate bits were made with), the
                // masked value back into the copy, and its exponent byte counted -- the select's first pass.  Two barriers and a pass
                // over the map less than gate bits | barrier | mask | barrier | pass 0: each is 1.2-1.7 us of dependent latency for a
                // router that shares its SIMDs with the VQ waves, and the routers are the fused launch's tail.
                // (hist[0] is clear: the coarse select's last pass did that, in front of its barrier.)
                unsigned int *h0 = sh->hist[0];
                RunCount rc;
                auto one = [&](int at, int parent) {
                    const float g = rd16(parent) < thr_c ? 1.0f : 0.0f;
                    const float m = l8w[at] * (1.0f - g);
                    l8w[at] = m;
                    rc.take(h0, f2key(m) >> 24);
                };
                if (rows2d) {
                    for (int y = wv; y < (int)h8; y += NWV)
                        for (int x = lane; x < w8i; x += 64) one(y * w8i + x, (y >> 1) * w16i + (x >> 1));
                } else {
                    for (int i = tid; i < (int)N8; i += NT) {
                        const int b = fdiv(i, mg_n8, n8i), r = i - b * n8i;
                        const int y = fdiv(r, mg_w8, w8i), x = r - y * w8i;
                        one(i, b * n16i + (y >> 1) * w16i + (x >> 1));
                    }
                }
                rc.finish_pass0(h0, lane);
                for (int i = tid; i < 256; i += NT) sh->hist[1][i] = 0;
                __syncthreads();
                thr_m = radix_select<NT, true>(rd8, N8, a.rank_m, sh, &si);
            } else {
            if (rows2d) {
                for (int y = wv; y < (int)h8; y += NWV)
                    for (int x = lane; x < w8i; x += 64) l8w[y * w8i + x] = l8w[y * w8i + x] * (1.0f - (gc_at(0, y, x) ? 1.0f : 0.0f));
            } else {
                for (int64_t i = tid; i < N8; i += NT) l8w[i] = l8w[i] * (1.0f - (gc_of8(i) ? 1.0f : 0.0f));
            }
            __syncthreads();
            thr_m = radix_select<NT>(rd8, N8, a.rank_m, sh, &si);        // (stage 2: the masked copy through the generic pointer)
            }
            if constexpr (ST) if (refine)       // (a gated element's 0 is exact: never re-evaluated, never overwritten)
                thr_m = refine_select<NT, 8, HELP, SPLIT, BAIL>(a, (int)(2 * seg + 1), nb, band, l8m, (int)N8, a.rank_m, thr_m,
                                             ExactGate{gc_bits, n8i, w8i, n16i, w16i, mg_n8, mg_w8}, seg * a.per, w8i, n8i, rs, sh, si, &bail, &refined);
            if (BAIL && bail) return 1;
        } else {
            thr_m = radix_select<NT>([&](int64_t i) { return e8[i] * (1.0f - (gc_of8(i) ? 1.0f : 0.0f)); }, N8, a.rank_m, sh);
        }
    }
    if (mode == 1) {      // :40-43
        SelInfo si;
        thr_m = radix_select<NT>(rd8, N8, a.rank_m, sh, &si);
        if constexpr (ST) if (refine)
            thr_m = refine_select<NT, 8, HELP, SPLIT, BAIL>(a, (int)(2 * seg + 1), nb, band, const_cast<float *>(e8), (int)N8, a.rank_m, thr_m, ExactGate{nullptr, 0, 0, 0, 0, 0, 0},
                                         seg * a.per, w8i, n8i, rs, sh, si, &bail, &refined);
        if (BAIL && bail) return 1;
    }
    auto gm_rule = [&](float v, bool gc) -> bool {
        switch (mode) {
        case 0: return (v < thr_m) && !gc;                  // :32
        case 1: return v < thr_m;                           // :44
        case 3: return !gc;                                 // :68
        case 5: return true;                                // :81
        default: return false;
        }
    };
    auto gm_of8 = [&](int64_t i) -> bool { return gm_rule(rd8(i), (mode == 0 || mode == 3) ? gc_of8(i) : false); };
    // past the last select: this workgroup no longer holds a band open (helpers of the launch may leave)
    unsigned int board_busy = 0;
    if (refine && a.rq.nq && tid == 0) {
        if (rs->busy_held) { rs->busy_held = 0; __hip_atomic_fetch_sub(a.rq.board + QB_BUSY, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (HELP) board_busy = ld_sc1(a.rq.board + QB_BUSY);          // (consumed behind the mask writes)
    }
    CGIC_STAMP(4);
    CGIC_RT_STAMP(2);
    if (rows2d) {
        for (int y = 2 * cy0 + wv; y < 2 * cy1; y += NWV)
            for (int x = lane; x < w8i; x += 64)
                mm[y * w8i + x] = gm_rule(rd8(y * w8i + x), (mode == 0 || mode == 3) ? gc_at(0, y, x) : false) ? 1 : 0;
    } else {
        for (int64_t i = tid; i < N8; i += NT) mm[i] = gm_of8(i) ? 1 : 0;
    }
    CGIC_STAMP(5);

    // ---- fine gate + optional gate tensor (:34,47,58,69,77,83,87,93): 4 consecutive x per thread
    // (w4 is a multiple of 4, so a quad never straddles a row, a medium pair or a coarse cell)
    float *gate = a.gate ? a.gate + seg * N4 * 3 : nullptr;
    const int W4 = (int)w4, W8 = (int)w8, W16 = (int)w16, NQ = (int)(N4 >> 2), n4i = (int)n4, qrow = W4 >> 2;
    auto fine_quad = [&](int b, int y, int x) {
        const int i = b * n4i + y * W4 + x;
        const int64_t c = (int64_t)b * n16 + (y >> 2) * W16 + (x >> 2);
        const bool gc = (gc_bits[c >> 6] >> (c & 63)) & 1ull;
        const int64_t m0 = (int64_t)b * n8 + (y >> 1) * W8 + (x >> 1);
        const bool gcm = (mode == 0 || mode == 3) ? gc : false;          // (the medium pair's coarse parent is this quad's)
        const bool gm0 = gm_rule(rd8(m0), gcm), gm1 = gm_rule(rd8(m0 + 1), gcm);
        bool gf0, gf1;
        switch (mode) {
        case 0: gf0 = !gc && !gm0; gf1 = !gc && !gm1; break;
        case 1: gf0 = !gm0; gf1 = !gm1; break;
        case 2: gf0 = gf1 = !gc; break;
        case 6: gf0 = gf1 = true; break;
        default: gf0 = gf1 = false; break;
        }
        *reinterpret_cast<int4 *>(mf + i) = make_int4(gf0, gf0, gf1, gf1);
        if (gate) {
            float *row = gate + ((int64_t)b * h4 + y) * 3 * w4;
            const float c1 = gc ? 1.f : 0.f, a0 = gm0 ? 1.f : 0.f, a1 = gm1 ? 1.f : 0.f;
            *reinterpret_cast<float4 *>(row + x) = make_float4(c1, c1, c1, c1);
            *reinterpret_cast<float4 *>(row + w4 + x) = make_float4(a0, a0, a1, a1);
            *reinterpret_cast<float4 *>(row + 2 * w4 + x) = make_float4(gf0 ? 1.f : 0.f, gf0 ? 1.f : 0.f, gf1 ? 1.f : 0.f, gf1 ? 1.f : 0.f);
        }
    };
    if (rows2d) {
        for (int y = 4 * cy0 + wv; y < 4 * cy1; y += NWV)
            for (int xq = lane; xq < qrow; xq += 64) fine_quad(0, y, xq << 2);
    } else {
        for (int q = tid; q < NQ; q += NT) {
            const int i = q << 2;
            const int b = fdiv(i, mg_n4, n4i), r = i - b * n4i;
            const int y = fdiv(r, mg_w4, W4), x = r - y * W4;
            fine_quad(b, y, x);
        }
    }
    CGIC_STAMP(6);
    CGIC_RT_STAMP(3);
    if (HELP && refine && a.rq.nq) {
        if (tid == 0) rs->flag = board_busy;
        __syncthreads();
        if (rs->flag) refine_help_while_busy(a, &rs->tl);
    }
    return refined ? 2 : 0;
}


// stage 1: e16 and e8 live in LDS (e8 is masked in place for the medium select); stage 2: only e8 (e16 is read from global
// memory by the coarse select's four passes); 0: nothing staged.  `budget`: the fused VQ + router launch keeps a router
// workgroup under half a CU's LDS so that it can share the CU with a VQ workgroup.  `refine`: room for RefineShared behind
// the maps (stage 1 only; *stage = -1 if that does not fit the budget).
__host__ __device__ inline size_t router_lds_bytes(int64_t N16, int64_t N8, int *stage, size_t budget = 96 * 1024, bool refine = false)
{
    size_t lds = kRouterSharedBytes + 8 * (size_t)((N16 + 63) / 64);
    int st = 0;
    if (refine) {
        const size_t need = lds + 4 * (size_t)(N16 + N8) + 16 + sizeof(RefineShared);
        if (need <= budget) { st = 1; lds = need; } else st = -1;
    } else if (lds + 4 * (size_t)(N16 + N8) <= budget) { st = 1; lds += 4 * (size_t)(N16 + N8); }
    else if (lds + 4 * (size_t)N8 <= budget) { st = 2; lds += 4 * (size_t)N8; }
    if (stage) *stage = st;
    return lds;
}

// host-side argument preparation shared by the stand-alone and the VQ-fused launch
int router_prepare(const float *e16, const float *e8, int64_t B, int64_t h16, int64_t w16, double c_ratio,
                   double m_ratio, int per_image, int32_t *mask_c, int32_t *mask_m, int32_t *mask_f, float *gate,
                   RouterArgs *out, int64_t *nseg, size_t *lds, size_t lds_budget = 96 * 1024, const cgic_pixels *refine = nullptr,
                   hipStream_t stream = nullptr, bool queues = false);

// segments whose maps do not fit the LDS budget below: refinement as a chain of launches over patched copies (cgic_router.hip)
bool router_refine_in_lds(int64_t B, int64_t h16, int64_t w16, int per_image);
int router_big(const float *e16, const float *e8, int64_t B, int64_t h16, int64_t w16, double c_ratio, double m_ratio, int per_image,
               int32_t *mask_c, int32_t *mask_m, int32_t *mask_f, float *gate, const cgic_pixels *refine, hipStream_t stream);

// LDS budget of a router workgroup in the fused VQ + router launch (two allocations per 160 KB CU); refinement is offered
// for segments that fit THIS budget, in the stand-alone launch too, so that one answer holds for both
constexpr size_t kRouterFusedLds = (size_t)78 * 1024;

}  // namespace cgic
