// cgic_decode_ss.hip -- the self-synchronising prefix decoder of the throughput mode (split off cgic_coder.hip in round 3).
#include "cgic_coder_dev.h"

namespace cgic {

// -------------------------------------------------------------------------------------------
// Self-synchronising decoder: ONE workgroup per image decodes all three index streams (round 2, second half).
//
// The kernels above find the codeword boundaries of a chunk for EVERY possible entry offset (64 speculative starts per
// 64-bit chunk, pointer doubling, function composition): worst-case parallel time, but ~11 000 wave instructions, 16 waves
// and 132 KB of LDS per 1 KB stream -- one workgroup per CU and stream for ~11 us.  With several batches in flight
// (pipeline.LaneStream) what counts is the CU time a launch consumes, not its latency, and that was the second largest
// item of the step (decode + merge 14.7 of 46.8 us).
//
// Here a lane owns a 64-bit chunk and simply GUESSES its entry offset (0), decodes the chunk serially from the LUT and
// notes where it ran out (exit offset = entry of the next chunk) and how many symbols it saw.  Then every lane compares
// its guess with its predecessor's exit and decodes again if they differ, until a sweep changes nothing.  The first chunk
// of a stream is right from the start, so chunk k is right after at most k sweeps (exact for any input: the fixpoint is
// unique); a Huffman stream re-synchronises within a few codewords, so in practice the second sweep already changes
// nothing for almost every chunk.  A block scan of the counts gives the output positions and a last walk stores the
// symbols.  Work: ~4 serial walks of ~9 LUT lookups per chunk instead of 64 starts x 6 doubling rounds; footprint: a
// 256-thread workgroup and LUT + stream bytes of LDS (41 KB for a 256x256 image), three or more workgroups per CU.
// Worst case (a stream built never to re-synchronise) degrades to one sweep per chunk -- still exact.
// Entry offsets stay below 64 because a codeword is at most 64 bits (max_len <= 64; longer tables take the serial path).
// -------------------------------------------------------------------------------------------
constexpr int kSsEnd = 0xFF;            // "the stream ended before this chunk" as an entry / exit offset

struct SsLayout {                       // wave-uniform description of the image's three streams
    int nbits[3];                       // payload bits
    int c[4];                           // first chunk of stream s; c[3] = total
    int off[3];                         // byte offset of the stream's copy in the LDS stage (16-byte aligned)
};
// (selects, not indexed loads: a private array indexed by a per-lane value would live in scratch memory)
__device__ __forceinline__ int sel3(int s, int v0, int v1, int v2) { return s == 0 ? v0 : (s == 1 ? v1 : v2); }

// 32 payload bits starting at bit `pos` (0..63) of the 96-bit big-endian window d0:d1:d2
__device__ __forceinline__ uint32_t ss_bits(uint32_t d0, uint32_t d1, uint32_t d2, int pos)
{
    const uint32_t hi = pos < 32 ? d0 : d1, lo = pos < 32 ? d1 : d2;
    const uint32_t r = __builtin_amdgcn_alignbit(hi, lo, 32u - ((uint32_t)pos & 31u));      // shift amount is mod 32
    return (pos & 31) ? r : hi;
}

// One serial walk over chunk g from `entry`: returns the exit offset (entry of the next chunk; kSsEnd when the stream
// ends in this chunk), *count = codewords that start in the chunk.  A lone wave issues an instruction every 5-9 cycles,
// so the walk is priced by its instruction count per codeword: 32-bit funnel shifts instead of 64-bit vector shifts, and
// the end-of-stream checks only in the stream's last two chunks (`rem` < 128).
template <bool WRITE, typename Put>
__device__ __forceinline__ int ss_walk(const TableDev &t, const uint32_t *lut, const uint8_t *stage, const SsLayout &L, int g,
                                       int entry, int *count, Put put)
{
    const int s = (g >= L.c[1]) + (g >= L.c[2]);
    const int ch = g - sel3(s, L.c[0], L.c[1], L.c[2]);
    const int rem = sel3(s, L.nbits[0], L.nbits[1], L.nbits[2]) - 64 * ch;            // payload bits from the start of this chunk
    *count = 0;
    if (entry == kSsEnd) return kSsEnd;
    // payload byte 8*ch sits at stage byte 1 + 8*ch (byte 0 is the pad count): five aligned words, shifted by one byte
    const uint32_t *q = reinterpret_cast<const uint32_t *>(stage + sel3(s, L.off[0], L.off[1], L.off[2]) + 8 * ch);
    const uint32_t r0 = __builtin_bswap32(q[0]), r1 = __builtin_bswap32(q[1]), r2 = __builtin_bswap32(q[2]),
                   r3 = __builtin_bswap32(q[3]), r4 = __builtin_bswap32(q[4]);
    const uint32_t d0 = __builtin_amdgcn_alignbit(r0, r1, 24), d1 = __builtin_amdgcn_alignbit(r1, r2, 24),
                   d2 = __builtin_amdgcn_alignbit(r2, r3, 24), d3 = __builtin_amdgcn_alignbit(r3, r4, 24);      // payload bits 0..127
    const int LB = t.lut_bits;
    int pos = entry, n = 0;
    if (t.max_len <= LB) {
        // The common case: every code fits the LUT window.  (w0:w1:w2) holds the 96 payload bits from `pos` on, left-aligned, and is
        // shifted by each code length with 32-bit funnel shifts (the 64-bit vector shifts of the first version are quarter-rate:
        // a lone wave is priced by its dependent instruction chain, ~12 VALU instructions + one LDS lookup per codeword now, ~32
        // before).  The loop is PREDICATED, not divergent: every lane runs the same instructions until no lane of the wave has a
        // codeword left (one wave-uniform branch per trip); as a structured loop with early exits it cost ~750 cycles per
        // codeword -- a lone wave pays ~20 cycles for every exec-mask update and taken branch.
        // End of walk without extra state: the staged LUT marks "no such prefix" with length 255 (decode_image_kernel), and
        // `remc` = min(rem, 128) bounds every legal end (a codeword that starts before bit 64 ends before bit 64 + 64): a lane
        // whose codeword is no code of the table, or ends beyond the stream, lands on pos > remc and is neither counted nor
        // continued.  pos < 64 after the loop: the stream ended at a codeword boundary inside this chunk.
        const uint32_t a0 = pos < 32 ? d0 : d1, a1 = pos < 32 ? d1 : d2, a2 = pos < 32 ? d2 : d3, a3 = pos < 32 ? d3 : 0u;
        const uint32_t sft = 32u - ((uint32_t)pos & 31u);                       // alignbit shifts right by (amount & 31)
        const bool al = ((uint32_t)pos & 31u) == 0u;
        uint32_t w0 = al ? a0 : __builtin_amdgcn_alignbit(a0, a1, sft), w1 = al ? a1 : __builtin_amdgcn_alignbit(a1, a2, sft),
                 w2 = al ? a2 : __builtin_amdgcn_alignbit(a2, a3, sft);
        const int remc = rem < 128 ? rem : 128, lim = rem < 64 ? rem : 64;
        const uint32_t ish = 30u - (uint32_t)LB;                                 // index in bytes: (w0 >> (32 - LB)) * 4
        const uint32_t imask = ((1u << LB) - 1u) << 2;
        const char *lutb = reinterpret_cast<const char *>(lut);
        auto step = [&]() {
            const bool go = pos < lim;
            const uint32_t e = *reinterpret_cast<const uint32_t *>(lutb + ((w0 >> ish) & imask));
            const int len = (int)(e & 0xFF);
            const int np = pos + len;
            const bool ok = go && np <= remc;                                   // a code of this table that ends inside the stream
            if (WRITE) { if (ok) put(s, n, (int)(e >> 8)); }
            const uint32_t r = 32u - (uint32_t)len;                             // (len 0 or 255 only on lanes that are done: their window is dead)
            w0 = __builtin_amdgcn_alignbit(w0, w1, r);
            w1 = __builtin_amdgcn_alignbit(w1, w2, r);
            w2 <<= (len & 31);
            pos = go ? np : pos;
            n += ok ? 1 : 0;
        };
        // two codewords per trip: the wave-uniform test and the taken branch cost a lone wave as much as three instructions
        while (__builtin_amdgcn_ballot_w64(pos < lim) != 0) {
            step();
            step();
        }
        *count = n;
        return (pos < 64 || pos > remc) ? kSsEnd : pos - 64;
    }
    while (pos < 64) {
        if (pos >= rem) { *count = n; return kSsEnd; }
        const uint32_t bits = ss_bits(d0, d1, d2, pos);
        const uint32_t e = lut[bits >> (32 - LB)];
        int len = (int)(e & 0xFF), sym = (int)(e >> 8);
        if (len == 0) {
            if (sym == 0xFFFFFF) { *count = n; return kSsEnd; }
            const unsigned long long w0 = ((unsigned long long)d0 << 32) | d1, w1 = ((unsigned long long)d2 << 32) | d3;
            const unsigned long long win = pos ? (w0 << pos) | (w1 >> (64 - pos)) : w0;
            int node = sym, k = LB;
            sym = -1;
            while (pos + k < rem && k < 64) {
                const int c = t.child[2 * node + (int)((win >> (63 - k)) & 1ull)];
                ++k;
                if (c == INT32_MIN) break;
                if (c < 0) { sym = ~c; break; }
                node = c;
            }
            if (sym < 0) { *count = n; return kSsEnd; }
            len = k;
        }
        if (pos + len > rem) { *count = n; return kSsEnd; }          // trailing partial codeword: dropped by the reference
        if (WRITE) put(s, n, sym);
        pos += len;
        ++n;
    }
    *count = n;
    return pos - 64;
}


// -------------------------------------------------------------------------------------------
// Round 4: ALL entry offsets at once instead of guessing.  A codeword that starts before bit 64 of a chunk ends at most
// max_len - 1 bits into the next one, so a chunk can only be entered at offsets 0 .. max_len - 1 (or not at all: the stream ended).
// For tables whose codes all fit the LUT window and max_len <= 15 (the Zipf-like table of a trained model: 13) a lane walks its
// chunk from EVERY such offset, the walks interleaved in one instruction stream: a single walk is a chain of dependent LUT
// lookups (~400 cycles per codeword for a lone wave), fifteen independent chains fill those gaps, so the pass costs about what
// three guessing walks cost.  The result is the chunk's entry -> exit FUNCTION (16 bytes: exit offset per entry, 15 = "stream
// ended"); functions compose (v_perm_b32 looks eight entries up at once), a wave scan + one cross-wave step give every chunk its
// true entry -- no fix-point sweeps (8.3 on the benchmark's streams, one per chunk on a stream built never to re-synchronise),
// no data-dependent running time.
// -------------------------------------------------------------------------------------------
constexpr int kSsMaxSweeps = 32;        // fix-point sweeps before the all-entries pass takes over (ordinary streams: 4-10, seen up to 16)
constexpr int kMeEntries = 15;          // entry offsets 0..14; code 15 = kMeEnd
constexpr uint32_t kMeEnd = 15;

struct MeFn { uint32_t r[4]; };         // byte e = exit code of entry e

__device__ __forceinline__ MeFn me_identity() { return MeFn{{0x03020100u, 0x07060504u, 0x0B0A0908u, 0x0F0E0D0Cu}}; }
__device__ __forceinline__ MeFn me_const0() { return MeFn{{0u, 0u, 0u, 0u}}; }      // every entry, "ended" included, -> 0: the next chunk starts a stream
// (cur o prev)[e] = cur[prev[e]]: prev's bytes are the selectors into cur's 16-byte table
__device__ __forceinline__ MeFn me_compose(const MeFn &prev, const MeFn &cur)
{
    MeFn o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sel = prev.r[k];
        const uint32_t s7 = sel & 0x07070707u;
        const uint32_t lo = __builtin_amdgcn_perm(cur.r[1], cur.r[0], s7);      // entries 0..7
        const uint32_t hi = __builtin_amdgcn_perm(cur.r[3], cur.r[2], s7);      // entries 8..15
        const uint32_t m = ((sel >> 3) & 0x01010101u) * 0xFFu;
        o.r[k] = (hi & m) | (lo & ~m);
    }
    return o;
}
__device__ __forceinline__ uint32_t me_apply(const MeFn &f, uint32_t e)
{
    const uint32_t w = e < 8 ? (e < 4 ? f.r[0] : f.r[1]) : (e < 12 ? f.r[2] : f.r[3]);
    return (w >> (8 * (e & 3))) & 0xFFu;
}
__device__ __forceinline__ MeFn me_shfl_up(const MeFn &f, int d)
{
    MeFn o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o.r[k] = (uint32_t)__shfl_up((int)f.r[k], d, kWave);
    return o;
}
// inclusive scan of function composition over the lanes of a wave (lane i: f_i o ... o f_0)
__device__ __forceinline__ MeFn me_wave_scan(MeFn f)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const MeFn p = me_shfl_up(f, d);
        const MeFn c = me_compose(p, f);
        if (lane >= d) f = c;
    }
    return f;
}

// The entry -> (exit, count) maps of chunk g for all entry offsets: returns the function; cnt_lo / cnt_hi get the codeword counts
// of entries 0..7 / 8..14, one byte each (a chunk holds at most 64 codewords).  Requires t.max_len <= t.lut_bits (every code in the
// staged LUT, "no such prefix" staged as length 255) and t.max_len <= 15.
__device__ __forceinline__ MeFn me_walk_all(const TableDev &t, const uint32_t *lut, const uint8_t *stage, const SsLayout &L, int g,
                                            unsigned long long *cnt_lo, unsigned long long *cnt_hi)
{
    const int s = (g >= L.c[1]) + (g >= L.c[2]);
    const int ch = g - sel3(s, L.c[0], L.c[1], L.c[2]);
    const int rem = sel3(s, L.nbits[0], L.nbits[1], L.nbits[2]) - 64 * ch;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(stage + sel3(s, L.off[0], L.off[1], L.off[2]) + 8 * ch);
    const uint32_t r0 = __builtin_bswap32(q[0]), r1 = __builtin_bswap32(q[1]), r2 = __builtin_bswap32(q[2]), r3 = __builtin_bswap32(q[3]);
    const uint32_t d0 = __builtin_amdgcn_alignbit(r0, r1, 24), d1 = __builtin_amdgcn_alignbit(r1, r2, 24), d2 = __builtin_amdgcn_alignbit(r2, r3, 24);
    const int LB = t.lut_bits;
    const int remc = rem < 128 ? rem : 128, lim = rem < 64 ? rem : 64;
    const uint32_t ish = 30u - (uint32_t)LB, imask = ((1u << LB) - 1u) << 2;
    const char *lutb = reinterpret_cast<const char *>(lut);
    const int E = t.max_len < kMeEntries ? t.max_len : kMeEntries;
    int pos[kMeEntries], n[kMeEntries];
#pragma unroll
    for (int e = 0; e < kMeEntries; ++e) { pos[e] = e < E ? e : 1 << 20; n[e] = 0; }       // (entries beyond max_len - 1 cannot occur: parked)
    for (;;) {
        bool any = false;
#pragma unroll
        for (int e = 0; e < kMeEntries; ++e) any |= pos[e] < lim;
        if (__builtin_amdgcn_ballot_w64(any) == 0) break;
#pragma unroll
        for (int e = 0; e < kMeEntries; ++e) {
            const int p = pos[e];
            const bool go = p < lim;
            const uint32_t bits = ss_bits(d0, d1, d2, p & 63);
            const uint32_t ent = *reinterpret_cast<const uint32_t *>(lutb + ((bits >> ish) & imask));
            const int np = p + (int)(ent & 0xFF);
            const bool ok = go && np <= remc;                       // a code of this table that ends inside the stream
            pos[e] = go ? np : p;
            n[e] += ok ? 1 : 0;
        }
    }
    MeFn f;
    unsigned long long lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) f.r[k] = 0;
#pragma unroll
    for (int e = 0; e < kMeEntries; ++e) {
        const int p = pos[e];
        const uint32_t x = (e >= E || p < 64 || p > remc) ? kMeEnd : (uint32_t)(p - 64);     // the stream ended inside (or before) this chunk
        f.r[e >> 2] |= x << (8 * (e & 3));
        if (e < 8) lo |= (unsigned long long)(uint32_t)n[e] << (8 * e);
        else hi |= (unsigned long long)(uint32_t)n[e] << (8 * (e - 8));
    }
    f.r[3] |= kMeEnd << 24;
    *cnt_lo = lo; *cnt_hi = hi;
    return f;
}

// only: -1 = this workgroup decodes the image's three index streams (the shipping form); 0..2 = that one stream (dev A/B of round 6:
// one workgroup per (image, stream), grid (B, 3) -- fewer sweeps per workgroup, three LUT stagings per image)
__device__ __forceinline__ void decode_image_body(const DecodeArgs &a, const int stage_cap, const int chunk_cap, const Blk blk, const int only = -1)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ int s_scan[kDecWaves + 1];
    __shared__ int s_base[4];
    __shared__ MeFn s_fn[kDecWaves];
    uint32_t *lut = sm;                                             // [1 << lut_bits]
    uint8_t *stage = reinterpret_cast<uint8_t *>(lut + (1 << a.tab.lut_bits));
    uint8_t *ent = stage + stage_cap, *ext = ent + chunk_cap, *cnt = ext + chunk_cap;
    const int tid = threadIdx.x, T = blockDim.x, lane = lane_id(), wave = tid >> 6, nw = T >> 6;
    const int64_t b = blk.x;
    CGIC_STAMP3(0);
    const uint8_t *in0 = a.in + (b * CGIC_NUM_STREAMS) * a.slot;
    int nb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) nb[s] = (a.stream_mask >> s & 1) ? a.nbytes[b * CGIC_NUM_STREAMS + s] : -2;
    const bool own_s[3] = {only < 0 || only == 0, only < 0 || only == 1, only < 0 || only == 2};
    int nb_all[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) { nb_all[s] = nb[s]; if (!own_s[s]) nb[s] = -2; }
    (void)nb_all;
    const int pad[3] = {in0[0], in0[a.slot], in0[2 * a.slot]};      // (slot memory is always readable)
    {   // LUT -> LDS: eight 16-byte loads in flight per thread (a plain strided loop pays one memory round trip per trip:
        // 32 trips of a 256-thread workgroup = 25 us)
        const uint4 *gl = reinterpret_cast<const uint4 *>(a.tab.lut);
        uint4 *dl = reinterpret_cast<uint4 *>(lut);
        const int n4 = (1 << a.tab.lut_bits) >> 2;
        // every code fits the LUT window (ss_walk's predicated loop): "no such prefix" (length 0) is staged as length 255, which
        // ends a walk without a test of its own
        const bool mark = a.tab.max_len <= a.tab.lut_bits;
        auto fix = [mark](uint32_t e) { return mark && (e & 0xFFu) == 0u ? e | 0xFFu : e; };
        if (n4 == 0 && tid < (1 << a.tab.lut_bits)) lut[tid] = fix(a.tab.lut[tid]);      // (a table of two symbols: a 2-entry LUT)
        for (int base = 0; base < n4; base += 8 * T) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = base + k * T + tid;
                v[k] = i < n4 ? gl[i] : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = base + k * T + tid;
                if (i < n4) dl[i] = uint4{fix(v[k].x), fix(v[k].y), fix(v[k].z), fix(v[k].w)};
            }
        }
    }
    const int64_t n_c = (a.h >> 2) * (a.w >> 2), n_m = (a.h >> 1) * (a.w >> 1), n_f = a.h * a.w;
    const int cap[3] = {(int)n_c, (int)n_m, (int)n_f};
    SsLayout L;
    int stage_used = 0;
    bool fits = true;
    L.c[0] = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (nb[s] > a.slot) nb[s] = (int)a.slot;                    // (a corrupt length cannot reach beyond the slot)
        int bits = nb[s] <= 0 || pad[s] == 0 ? 0 : (nb[s] - 1) * 8 - pad[s];        // remove_padding :131-138; text[:-0] is empty
        bits = bits < 0 ? 0 : bits;
        L.nbits[s] = bits;
        L.c[s + 1] = L.c[s] + ((bits + 63) >> 6);
        L.off[s] = stage_used;
        stage_used += bits ? (((bits + 7) >> 3) + 1 + 24 + 15) & ~15 : 0;          // header + payload + the two words a walk reads past its chunk
    }
    const int C = L.c[3];
    if (stage_used > stage_cap || C > chunk_cap) fits = false;     // more bits than the grids can hold symbols: overflow
    if (tid == 0) {
        if (a.status) a.status[b] = 0;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (!own_s[s]) continue;
            if (nb[s] <= 0) a.dcount[b * 3 + s] = nb[s] == 0 ? -1 : -2;             // empty file (None) / not sent
            else if (!fits) a.dcount[b * 3 + s] = -3;
        }
    }
    if (!fits) return;
    CGIC_STAMP3(1);
    // stage the stream bytes (16-byte copies; the tail beyond the stream is never interpreted: every walk checks `rem`)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (!L.nbits[s]) continue;
        const int words = ((((L.nbits[s] + 7) >> 3) + 1 + 24 + 15) & ~15) >> 4;
        const uint4 *g = reinterpret_cast<const uint4 *>(in0 + s * a.slot);
        uint4 *d = reinterpret_cast<uint4 *>(stage + L.off[s]);
        const int lim = (int)(a.slot >> 4);                         // stay inside the slot
        for (int i = tid; i < words; i += T) d[i] = i < lim ? g[i] : uint4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    CGIC_STAMP3(2);
    // blocked ownership: lane tid owns chunks [g0, g1)
    const int R = (C + T - 1) / T;
    const int g0 = tid * R < C ? tid * R : C, g1 = g0 + R < C ? g0 + R : C;
    auto is_first = [&](int g) { return g == L.c[0] || g == L.c[1] || g == L.c[2]; };
    auto nop = [](int, int, int) {};
    int dbg_sweeps = 0;
    const bool all_entries = a.tab.max_len <= a.tab.lut_bits && a.tab.max_len <= kMeEntries;       // (wave-uniform)
    // Guessing first: on ordinary streams it converges in a handful of sweeps and costs less than walking every entry offset
    // (measured, 64 images of 256x256: 30.3 vs 32.2 us decode + merge; all-fine grids 33.7 vs 57.1).  A stream that does not
    // re-synchronise -- one 13-bit codeword repeated: one sweep per chunk, 207 sweeps, 852 us -- is cut off after kSsMaxSweeps
    // and finished by the all-entries pass, whose running time does not depend on the data (56 us for that batch).
    bool stuck = false;
    {
        int prev = 0;
        for (int g = g0; g < g1; ++g) {
            // the guess: the smallest offset in the residue class the code lengths allow (0 when their gcd is 1); inside the
            // lane's own run the predecessor is known
            int e = prev;
            if (is_first(g)) e = 0;
            else if (g == g0) {
                const int sg = (g >= L.c[1]) + (g >= L.c[2]);
                const int chg = g - sel3(sg, L.c[0], L.c[1], L.c[2]);
                const int m = (64 * chg) % a.tab.len_gcd;
                e = m ? a.tab.len_gcd - m : 0;
            }
            int n;
            prev = ss_walk<false>(a.tab, lut, stage, L, g, e, &n, nop);
            ent[g] = (uint8_t)e; ext[g] = (uint8_t)prev; cnt[g] = (uint8_t)n;
        }
    }
    __syncthreads();
    CGIC_STAMP3(3);
    for (;;) {
        ++dbg_sweeps;
        int changed = 0;
        for (int g = g0; g < g1; ++g) {
            const int e = is_first(g) ? 0 : ext[g - 1];
            if (e != ent[g]) {
                int n;
                const int x = ss_walk<false>(a.tab, lut, stage, L, g, e, &n, nop);
                ent[g] = (uint8_t)e; ext[g] = (uint8_t)x; cnt[g] = (uint8_t)n;
                changed = 1;
            }
        }
        if (!__syncthreads_or(changed)) break;
        if (all_entries && dbg_sweeps >= kSsMaxSweeps) { stuck = true; break; }
    }
    if (stuck) {
        // ---- every entry offset at once; the chunks' entry -> exit functions composed by a scan
        MeFn mine = me_identity();
        unsigned long long c_lo = 0, c_hi = 0;
        for (int g = g0; g < g1; ++g) {
            MeFn f = me_walk_all(a.tab, lut, stage, L, g, &c_lo, &c_hi);
            if (is_first(g + 1)) f = me_const0();                 // whatever this chunk's exit is, the next chunk starts a stream at 0
            mine = me_compose(mine, f);
        }
        const MeFn incl = me_wave_scan(mine);
        if (lane == kWave - 1) s_fn[wave] = incl;
        __syncthreads();
        // (few waves: every lane composes the totals of the waves before its own)
        MeFn before = me_identity();
        for (int w2 = 0; w2 < wave; ++w2) before = me_compose(before, s_fn[w2]);
        MeFn prev = me_shfl_up(incl, 1);
        if (lane == 0) prev = me_identity();
        uint32_t e = me_apply(prev, me_apply(before, 0u));         // this lane's first chunk is entered here (0 at a stream start by construction)
        if (g0 < g1 && is_first(g0)) e = 0;
        if (R == 1) {
            if (g0 < g1) {
                const unsigned long long cw = e < 8 ? c_lo : c_hi;
                ent[g0] = (uint8_t)(e == kMeEnd ? kSsEnd : (int)e);
                cnt[g0] = e == kMeEnd ? (uint8_t)0 : (uint8_t)((cw >> (8 * (e & 7))) & 0xFFull);
            }
        } else {
            int en = e == kMeEnd ? kSsEnd : (int)e;
            for (int g = g0; g < g1; ++g) {
                if (is_first(g)) en = 0;
                int n;
                const int x = ss_walk<false>(a.tab, lut, stage, L, g, en, &n, nop);
                ent[g] = (uint8_t)en; cnt[g] = (uint8_t)n;
                en = x;
            }
        }
        dbg_sweeps += 1;
        __syncthreads();
    }
    CGIC_STAMP3(4);
    if (a.stats && tid == 0) {
        atomicAdd(&a.stats[0], (unsigned int)dbg_sweeps);
        atomicAdd(&a.stats[1], 1u);
        atomicMax(&a.stats[2], (unsigned int)dbg_sweeps);
    }
#ifdef CGIC_PHASE_CLOCKS
    if (blk.x == 0 && tid == 0) g_phase_clk[9] = dbg_sweeps;
#endif
    // output positions: exclusive prefix of the counts over the chunks, restarted at every stream
    int mine = 0;
    for (int g = g0; g < g1; ++g) mine += cnt[g];
    int inc = wave_inclusive_scan(mine);
    if (lane == kWave - 1) s_scan[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const int v = lane < nw ? s_scan[lane] : 0;
        const int vi = wave_inclusive_scan(v);
        if (lane < nw) s_scan[lane] = vi - v;
        if (lane == nw - 1) s_scan[kDecWaves] = vi;
    }
    __syncthreads();
    int run = s_scan[wave] + inc - mine;
    const int total = s_scan[kDecWaves];
    {
        int r = run;
        for (int g = g0; g < g1; ++g) {
            if (g == L.c[1]) s_base[1] = r;
            if (g == L.c[2]) s_base[2] = r;
            r += cnt[g];
        }
        if (tid == 0) {
            s_base[0] = 0; s_base[3] = total;
            if (L.c[2] == C) s_base[2] = total;
            if (L.c[1] == C) s_base[1] = total;
        }
    }
    __syncthreads();
    CGIC_STAMP3(5);
    uint16_t *dst = a.dsym + b * (n_c + n_m + n_f);
    for (int g = g0; g < g1; ++g) {
        const int s = (g >= L.c[1]) + (g >= L.c[2]);
        const int at = run - s_base[s];
        const int cap_s = sel3(s, cap[0], cap[1], cap[2]);
        uint16_t *dst_s = dst + sel3(s, 0, (int)n_c, (int)(n_c + n_m)) + at;
        int n;
        ss_walk<true>(a.tab, lut, stage, L, g, (int)ent[g], &n,
                      [&](int, int k, int sym) { if (at + k < cap_s) dst_s[k] = (uint16_t)sym; });
        run += n;
    }
    CGIC_STAMP3(6);
    if (tid < 3 && (only < 0 || only == tid) && sel3(tid, nb[0], nb[1], nb[2]) > 0) {
        const int n = s_base[tid + 1] - s_base[tid];
        a.dcount[b * 3 + tid] = n > sel3(tid, cap[0], cap[1], cap[2]) ? -3 : n;
    }
}

__global__ __launch_bounds__(kDecThreads) void decode_image_kernel(DecodeArgs a, int stage_cap, int chunk_cap)
{
    decode_image_body(a, stage_cap, chunk_cap, own_blk());
}
// dev A/B (CGIC_SS_PER_STREAM=1): grid (B, 3), blockIdx.y = the stream
__global__ __launch_bounds__(kDecThreads) void decode_image_stream_kernel(DecodeArgs a, int stage_cap, int chunk_cap)
{
    decode_image_body(a, stage_cap, chunk_cap, own_blk(), (int)blockIdx.y);
}

// several shape groups in one launch (cgic_common.h: launch groups); the groups share the workgroup size (checked by
// cgic_group_launch) and take the largest LDS footprint
__global__ __launch_bounds__(kDecThreads) void decode_image_grouped_kernel(Grouped<DecodeImageArgs> g)
{
    Blk blk;
    const DecodeImageArgs &p = g.a[group_locate(g, &blk)];
    decode_image_body(p.a, p.stage_cap, p.chunk_cap, blk);
}

}  // namespace cgic

using namespace cgic;
static int decode_image_grouped_launch(const GroupRec *const *recs, int n, hipStream_t s)
{
    Grouped<DecodeImageArgs> g;
    size_t lds;
    int rc = fill_grouped(recs, n, &g, &lds);
    if (rc) return rc;
    rc = ensure_dynamic_lds((const void *)decode_image_grouped_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(decode_image_grouped_kernel, dim3(g.start[kMaxGroups]), recs[0]->block, lds, s, g);
    return launch_check("decode_image_grouped_kernel");
}
static GroupedRegistrar reg_decode_image(KID_DECODE_IMAGE, decode_image_grouped_launch);
